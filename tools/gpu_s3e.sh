cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fp8w_gpu.py -m gpu -x -q > $D/tests_fp8.log 2>&1; echo "fp8w tests rc=$?"; tail -n 12 $D/tests_fp8.log | cut -c1-300
timeout 300 python bench.py --steps 3 --warmup 1 --cpu-frames 0 --dtype fp8w > $D/bench_b1_fp8w.log 2>&1; echo "b1 fp8w rc=$?"; tail -n 1 $D/bench_b1_fp8w.log | cut -c1-120; tail -n 1 $D/bench_b1_fp8w.log | grep -o '"phase_ms[^}]*}'; tail -n 1 $D/bench_b1_fp8w.log | grep -o '"roofline.*"frac": [0-9.]*'
timeout 300 python bench.py --steps 3 --warmup 1 --cpu-frames 0 > $D/bench_b1.log 2>&1; echo "b1 bf16 rc=$?"; tail -n 1 $D/bench_b1.log | cut -c1-120; tail -n 1 $D/bench_b1.log | grep -o '"phase_ms[^}]*}'
timeout 300 python bench.py --steps 3 --warmup 1 --cpu-frames 0 --opt glds_w8=1 > $D/bench_b1_w8.log 2>&1; echo "b1 glds_w8 rc=$?"; tail -n 1 $D/bench_b1_w8.log | grep -o '"phase_ms[^}]*}'
timeout 300 python bench.py --batch 64 --steps 1 --warmup 1 --cpu-frames 0 --dtype fp8w > $D/bench_b64_fp8w.log 2>&1; echo "b64 fp8w rc=$?"; tail -n 1 $D/bench_b64_fp8w.log | cut -c1-120; tail -n 1 $D/bench_b64_fp8w.log | grep -o '"phase_ms[^}]*}'
timeout 300 python bench.py --batch 8 --steps 1 --warmup 1 --cpu-frames 0 --dtype fp8w > $D/bench_b8_fp8w.log 2>&1; echo "b8 fp8w rc=$?"; tail -n 1 $D/bench_b8_fp8w.log | cut -c1-120
timeout 900 python -m pytest tests -m gpu -x -q > $D/tests.log 2>&1; echo "tests rc=$?"; tail -n 5 $D/tests.log
