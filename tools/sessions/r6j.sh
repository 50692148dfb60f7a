cd $GRAFT_REPO_ROOT
D=gpurun_out/r6j; mkdir -p $D
export TMPDIR=/tmp
timeout 300 python tools/fresh_box_probe.py --out $D/first > $D/probe.log 2>&1; echo "probe rc=$?"
for b in 2 3; do for o in 0 1 2; do
  timeout 300 python bench.py --batch $b --no-side --cpu-frames 0 --steps 3 --warmup 1 --opt persist=0 --opt nsplit=$o > $D/bench_b${b}_$o.log 2>&1
  echo "batch $b nsplit=$o: $(tail -n 1 $D/bench_b${b}_$o.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["phase_ms"], d["roofline"]["step_us"])' 2>&1 | tail -n 1)"
done; done
