cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["phase_ms"], d["roofline"]["launch_us"], d["roofline"]["frac"])'
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "split_k or m_split or linear" > $D/t_ops.log 2>&1; echo "ops rc=$?"; tail -n 2 $D/t_ops.log
timeout 300 python bench.py --batch 64 --steps 2 --warmup 1 --cpu-frames 0 --no-side > $D/b64.log 2>&1; tail -n 1 $D/b64.log | python -c "$P" b64
timeout 300 python bench.py --batch 64 --steps 2 --warmup 1 --cpu-frames 0 --no-side --opt gs_formal=1 > $D/b64f.log 2>&1; tail -n 1 $D/b64f.log | python -c "$P" b64_formal
timeout 300 python bench.py --batch 8 --steps 3 --warmup 1 --cpu-frames 0 --no-side > $D/b8.log 2>&1; tail -n 1 $D/b8.log | python -c "$P" b8
timeout 600 python tools/ktrace_step.py --out $D/ktrace_b64 --spg 8 --batch 64 > $D/ktrace_b64.log 2>&1; head -6 $D/ktrace_b64_timeline.csv
