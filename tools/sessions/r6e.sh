# round 6, session e: the whole GPU suite on the round's tree; batch 2 / 3 / 4 on the launch chain; kernel breakdown of the fp32 mode
cd $GRAFT_REPO_ROOT
D=gpurun_out/r6e; mkdir -p $D
export TMPDIR=/tmp
timeout 300 python tools/fresh_box_probe.py --out $D/first > $D/probe.log 2>&1; echo "probe rc=$?"
timeout 2400 python -m pytest tests -m gpu -x -q > $D/tests.log 2>&1; echo "tests rc=$?"; tail -n 4 $D/tests.log
for b in 2 3 4 8; do
  timeout 300 python bench.py --batch $b --no-side --cpu-frames 0 --steps 4 --warmup 1 > $D/bench_b$b.log 2>&1
  echo "batch $b: $(tail -n 1 $D/bench_b$b.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["phase_ms"], d["roofline"]["step_us"])' 2>&1 | tail -n 1)"
done
(cd /tmp && rm -rf /tmp/prof32 && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof32 -o f32 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --dtype fp32 --steps 1 --warmup 1 --cpu-frames 0 --no-side > $GRAFT_REPO_ROOT/$D/prof32.log 2>&1); echo "prof32 rc=$?"
cp /tmp/prof32/f32_kernel_stats.csv $D/ 2>/dev/null; head -n 14 $D/f32_kernel_stats.csv | cut -c1-150
