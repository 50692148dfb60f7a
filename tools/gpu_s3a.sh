cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_modules_gpu.py tests/test_ops_gpu.py -m gpu -x -q -k "modules or split_k or linear_gemm" > $D/tests_new.log 2>&1; echo "new tests rc=$?"; tail -n 6 $D/tests_new.log
for t in 1 256 512 1024; do
  timeout 300 python bench.py --batch 64 --steps 1 --warmup 1 --cpu-frames 0 --opt gs_target_wgs=$t > $D/bench_b64_t$t.log 2>&1; echo "b64 target=$t rc=$?"; tail -n 1 $D/bench_b64_t$t.log | cut -c1-120; tail -n 1 $D/bench_b64_t$t.log | grep -o '"phase_ms[^}]*}'
done
timeout 300 python bench.py --batch 64 --steps 1 --warmup 1 --cpu-frames 0 --profile-kernels 64 > $D/bench_b64_prof.log 2>&1; tail -n 1 $D/bench_b64_prof.log | grep -o '"kernel_us[^}]*}'
timeout 300 python bench.py --batch 8 --steps 1 --warmup 1 --cpu-frames 0 > $D/bench_b8.log 2>&1; echo "b8 rc=$?"; tail -n 1 $D/bench_b8.log | cut -c1-120
timeout 900 python -m pytest tests -m gpu -x -q > $D/tests.log 2>&1; echo "tests rc=$?"; tail -n 5 $D/tests.log
