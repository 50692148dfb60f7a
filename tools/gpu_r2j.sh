# round 2, call J: attention v2 -- parity, microbench, end-to-end lines
cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_modules_gpu.py -q -x -k "attention or encoder or Multihead or block" > $D/tests_attn.log 2>&1; echo "attn tests rc=$?"; tail -n 8 $D/tests_attn.log
timeout 300 python tools/attn_bench.py > $D/attn_bench.log 2>&1; cat $D/attn_bench.log | tail -n 6
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_parity_sizes_gpu.py -q -x > $D/tests_engine.log 2>&1; echo "engine tests rc=$?"; tail -n 4 $D/tests_engine.log
timeout 300 python bench.py --steps 2 --warmup 1 --cpu-frames 0 --batch 64 > $D/bench_b64.log 2>&1; echo "b64 $(tail -n 1 $D/bench_b64.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["phase_ms"], d["roofline"]["launch_us"], d["roofline"]["frac"])')"
timeout 300 python bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-c3 > $D/bench_b1.log 2>&1; echo "b1 $(tail -n 1 $D/bench_b1.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["phase_ms"], d["roofline"]["launch_us"], d["roofline"]["frac"])')"
