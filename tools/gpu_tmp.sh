cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_fp8w_gpu.py tests/test_serving_gpu.py -m gpu -x -q > $D/tests_engine.log 2>&1; echo "engine tests rc=$?"; tail -n 6 $D/tests_engine.log | cut -c1-300
for o in "gs_xf=1" "gs_xf=0"; do
  timeout 300 python bench.py --batch 64 --steps 1 --warmup 1 --cpu-frames 0 --opt $o > $D/bench_b64_$o.log 2>&1; echo "b64 $o rc=$?"; tail -n 1 $D/bench_b64_$o.log | grep -o '"value": [0-9.]*\|"phase_ms[^}]*}'
done
timeout 300 python bench.py --batch 8 --steps 1 --warmup 1 --cpu-frames 0 > $D/bench_b8.log 2>&1; echo "b8 rc=$?"; tail -n 1 $D/bench_b8.log | grep -o '"value": [0-9.]*\|"phase_ms[^}]*}'
timeout 300 python bench.py --batch 64 --steps 1 --warmup 1 --cpu-frames 0 --dtype fp8w > $D/bench_b64_fp8w.log 2>&1; echo "b64 fp8w rc=$?"; tail -n 1 $D/bench_b64_fp8w.log | grep -o '"value": [0-9.]*\|"phase_ms[^}]*}'
