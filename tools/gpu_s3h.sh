cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_serving_gpu.py -m gpu -x -q > $D/tests_serving.log 2>&1; echo "serving tests rc=$?"; tail -n 15 $D/tests_serving.log | cut -c1-300
timeout 900 python -m pytest tests -m gpu -x -q > $D/tests.log 2>&1; echo "tests rc=$?"; tail -n 4 $D/tests.log
timeout 600 python tools/serve_bench.py --n 192 --max-batch 64 > $D/serve_bench.log 2>&1; echo "serve bench rc=$?"; grep -v amdgpu $D/serve_bench.log | tail -5
timeout 300 python bench.py --batch 64 --steps 1 --warmup 1 --cpu-frames 0 > $D/bench_b64.log 2>&1; echo "b64 rc=$?"; tail -n 1 $D/bench_b64.log | grep -o '"value": [0-9.]*\|"phase_ms[^}]*}'
