# round 2, call C: forward tests, per-wave ktrace timeline, hand-off edge microbenchmark, rpw probes
cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_forward_gpu.py -q -x > $D/tests_fwd.log 2>&1; echo "fwd tests rc=$?"; tail -n 5 $D/tests_fwd.log
timeout 120 tools/bin/ubench_edges > $D/ubench_edges.json 2> $D/ubench_edges.err; echo "edges rc=$?"; cat $D/ubench_edges.json; tail -n 3 $D/ubench_edges.err
timeout 600 python tools/ktrace_step.py --out $D/ktrace --spg 8 > $D/ktrace.log 2>&1; echo "ktrace rc=$?"; tail -n 2 $D/ktrace.log | cut -c1-2500
for R in 1 2 4; do timeout 300 python bench.py --steps 4 --warmup 2 --cpu-frames 0 --no-c3 --opt gemv1_rpw=$R > $D/bench_rpw$R.log 2>&1; echo "rpw=$R $(tail -n 1 $D/bench_rpw$R.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["phase_ms"], d["roofline"]["launch_us"])')"; done
