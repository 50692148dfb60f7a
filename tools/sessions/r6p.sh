cd $GRAFT_REPO_ROOT
D=gpurun_out/r6p; mkdir -p $D
export TMPDIR=/tmp
timeout 300 python tools/fresh_box_probe.py --out $D/first > $D/probe.log 2>&1; echo "probe rc=$?"
timeout 1500 python -m pytest tests -m gpu -x -q > $D/tests.log 2>&1; echo "gpu tests rc=$?"; tail -n 12 $D/tests.log | cut -c1-400
for b in 2 3 4; do timeout 300 python bench.py --batch $b --no-side --cpu-frames 0 --steps 4 --warmup 1 > $D/bench_b$b.log 2>&1; echo "batch $b: $(tail -n 1 $D/bench_b$b.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["phase_ms"], d["roofline"]["step_us"], d["roofline"]["frac"], d["config"].get("persist"))' 2>&1 | tail -n 1)"; done
