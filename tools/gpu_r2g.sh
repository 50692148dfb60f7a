# round 2, call G: gemm_skinny rework (rotated X traversal, single burst, prefetched epilogue operands)
cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_fp8w_gpu.py tests/test_serving_gpu.py tests/test_ops_gpu.py -q -x > $D/tests.log 2>&1; echo "tests rc=$?"; tail -n 5 $D/tests.log
timeout 300 python bench.py --steps 2 --warmup 1 --cpu-frames 0 --batch 64 > $D/bench_b64.log 2>&1; echo "b64 $(tail -n 1 $D/bench_b64.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["phase_ms"], d["roofline"]["launch_us"], d["roofline"]["frac"])')"
timeout 300 python bench.py --steps 2 --warmup 1 --cpu-frames 0 --batch 8 > $D/bench_b8.log 2>&1; echo "b8 $(tail -n 1 $D/bench_b8.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["phase_ms"], d["roofline"]["launch_us"])')"
timeout 600 python bench.py --steps 2 --warmup 1 --cpu-frames 0 --d-model 1536 --layers 24 --nhead 16 --dtype fp8 --batch 32 > $D/bench_c5.log 2>&1; echo "c5 fp8 $(tail -n 1 $D/bench_c5.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["phase_ms"], d["roofline"]["launch_us"], d["roofline"]["frac"])')"
timeout 600 python tools/ktrace_step.py --out $D/ktrace_b64 --spg 8 --batch 64 > $D/ktrace_b64.log 2>&1; echo "ktrace b64 rc=$?"; head -n 7 $D/ktrace_b64_timeline.csv
timeout 600 python tools/ktrace_step.py --out $D/ktrace_c5 --spg 8 --batch 32 --d-model 1536 --layers 12 --dtype fp8w > $D/ktrace_c5.log 2>&1; echo "ktrace c5 rc=$?"; head -n 7 $D/ktrace_c5_timeline.csv
