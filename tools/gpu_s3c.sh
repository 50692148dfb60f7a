cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "variants or ln_linear or split_k" > $D/tests_new.log 2>&1; echo "new tests rc=$?"; tail -n 4 $D/tests_new.log
timeout 600 python tools/gs_bench.py > $D/gs_bench.log 2>&1; echo "gs_bench rc=$?"; cat $D/gs_bench.log | grep -v amdgpu.ids
for o in "gs_variant=1" "gs_fuse_ln=0" "gs_fuse_ln=1" "gs_wn=1" "gs_wn=2"; do
  timeout 300 python bench.py --batch 64 --steps 1 --warmup 1 --cpu-frames 0 --opt $o --profile-kernels 64 > $D/bench_b64_$o.log 2>&1; echo "b64 $o rc=$?"; tail -n 1 $D/bench_b64_$o.log | cut -c1-120; tail -n 1 $D/bench_b64_$o.log | grep -o '"phase_ms[^}]*}'; tail -n 1 $D/bench_b64_$o.log | grep -o '"kernel_us[^}]*}'
done
timeout 300 python bench.py --batch 8 --steps 1 --warmup 1 --cpu-frames 0 > $D/bench_b8.log 2>&1; echo "b8 rc=$?"; tail -n 1 $D/bench_b8.log | cut -c1-120
timeout 300 python bench.py --steps 3 --warmup 1 --cpu-frames 0 --profile-kernels 64 > $D/bench_b1.log 2>&1; echo "b1 rc=$?"; tail -n 1 $D/bench_b1.log | cut -c1-120;  tail -n 1 $D/bench_b1.log | grep -o '"phase_ms[^}]*}'; tail -n 1 $D/bench_b1.log | grep -o '"kernel_us[^}]*}'
timeout 300 python tools/op_chain_bench.py > $D/op_chain.log 2>&1; grep -v amdgpu.ids $D/op_chain.log | head -12
timeout 900 python -m pytest tests -m gpu -x -q > $D/tests.log 2>&1; echo "tests rc=$?"; tail -n 5 $D/tests.log
