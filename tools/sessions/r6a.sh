# round 6, session a: fresh-box probe first; the pruned persistent forms; small-allocation arena A/B on the headline
cd $GRAFT_REPO_ROOT
D=gpurun_out/r6a; mkdir -p $D
export TMPDIR=/tmp
timeout 300 python tools/fresh_box_probe.py --out $D/first > $D/probe.log 2>&1; echo "probe rc=$?"; tail -n 2 $D/probe.log | cut -c1-300
timeout 900 python -m pytest tests/test_persist_gpu.py tests/test_eos_gpu.py -x -q > $D/tests_persist.log 2>&1; echo "persist tests rc=$?"; tail -n 3 $D/tests_persist.log
for i in 1 2; do
  for a in 0 1; do
    VLE_ARENA=$a timeout 300 python bench.py --no-side --cpu-frames 0 --steps 10 --warmup 3 > $D/bench_arena${a}_$i.log 2>&1
    echo "arena=$a run $i: $(tail -n 1 $D/bench_arena${a}_$i.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["phase_ms"], d["roofline"]["step_us"])' 2>&1 | tail -n 1)"
  done
done
