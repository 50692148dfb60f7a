cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $D/tests.log 2>&1; echo "tests rc=$?"; tail -n 4 $D/tests.log
timeout 300 python bench.py --batch 64 --steps 2 --warmup 1 --cpu-frames 0 > $D/bench_b64.log 2>&1; echo "b64 rc=$?"; tail -n 1 $D/bench_b64.log | grep -o '"value": [0-9.]*\|"phase_ms[^}]*}'
timeout 300 python bench.py --batch 8 --steps 2 --warmup 1 --cpu-frames 0 > $D/bench_b8.log 2>&1; echo "b8 rc=$?"; tail -n 1 $D/bench_b8.log | grep -o '"value": [0-9.]*\|"phase_ms[^}]*}'
timeout 300 python bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-c3 > $D/bench_b1.log 2>&1; echo "b1 rc=$?"; tail -n 1 $D/bench_b1.log | grep -o '"value": [0-9.]*\|"phase_ms[^}]*}'
timeout 600 python tools/serve_bench.py --n 192 --max-batch 64 --harvest-min 8 16 > $D/serve_bench.log 2>&1; echo "serve rc=$?"; grep -v amdgpu $D/serve_bench.log | tail -4
