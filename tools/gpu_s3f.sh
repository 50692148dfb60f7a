cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fp8w_gpu.py -m gpu -q > $D/tests_fp8.log 2>&1; echo "fp8w tests rc=$?"; tail -n 12 $D/tests_fp8.log | cut -c1-300
timeout 900 python -m pytest tests -m gpu -x -q > $D/tests.log 2>&1; echo "tests rc=$?"; tail -n 5 $D/tests.log
timeout 600 python bench.py > $D/bench_default.log 2>&1; echo "default bench rc=$?"; tail -n 1 $D/bench_default.log
timeout 300 python bench.py --batch 64 --steps 1 --warmup 1 --cpu-frames 0 --opt glds_w8=0 > $D/bench_b64_w80.log 2>&1; echo "b64 w8=0 rc=$?"; tail -n 1 $D/bench_b64_w80.log | grep -o '"phase_ms[^}]*}'
python -c "import __graft_entry__ as g; g.smoke()" > $D/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 3 $D/smoke.log
