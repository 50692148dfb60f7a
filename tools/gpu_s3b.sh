cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_modules_gpu.py -m gpu -q > $D/tests_mod.log 2>&1; echo "modules tests rc=$?"; tail -n 6 $D/tests_mod.log
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "split_k or linear_gemm" > $D/tests_new.log 2>&1; echo "splitk tests rc=$?"; tail -n 4 $D/tests_new.log
for t in 1 256 512; do
  timeout 300 python bench.py --batch 64 --steps 1 --warmup 1 --cpu-frames 0 --opt gs_target_wgs=$t --profile-kernels 64 > $D/bench_b64_t$t.log 2>&1; echo "b64 target=$t rc=$?"; tail -n 1 $D/bench_b64_t$t.log | cut -c1-120; tail -n 1 $D/bench_b64_t$t.log | grep -o '"phase_ms[^}]*}'; tail -n 1 $D/bench_b64_t$t.log | grep -o '"kernel_us[^}]*}'
done
for t in 0 4 8 12; do
  timeout 300 python bench.py --steps 2 --warmup 1 --cpu-frames 0 --opt temporal_layers=$t > $D/bench_b1_tl$t.log 2>&1; echo "b1 temporal_layers=$t rc=$?"; tail -n 1 $D/bench_b1_tl$t.log | grep -o '"phase_ms[^}]*}'
done
timeout 600 python tools/chains_bench.py --batch 64 --chains 1 2 4 --opt gs_target_wgs=1 > $D/chains64.log 2>&1; echo "chains rc=$?"; cat $D/chains64.log | tail -5
timeout 600 python tools/chains_bench.py --batch 8 --chains 1 2 4 8 --opt gs_target_wgs=1 > $D/chains8.log 2>&1; echo "chains rc=$?"; cat $D/chains8.log | tail -5
timeout 900 python -m pytest tests -m gpu -x -q > $D/tests.log 2>&1; echo "tests rc=$?"; tail -n 5 $D/tests.log
