cd $GRAFT_REPO_ROOT
D=gpurun_out/r6m; mkdir -p $D
export TMPDIR=/tmp
timeout 300 python tools/fresh_box_probe.py --out $D/first > $D/probe.log 2>&1; echo "probe rc=$?"
timeout 900 python -m pytest tests/test_persist_gpu.py -x -q -k "batched or two_utterances or default_where" > $D/tests_nb.log 2>&1; echo "nb tests rc=$?"; tail -n 25 $D/tests_nb.log
for b in 2 3 4; do timeout 300 python bench.py --batch $b --no-side --cpu-frames 0 --steps 3 --warmup 1 > $D/bench_b$b.log 2>&1; echo "batch $b: $(tail -n 1 $D/bench_b$b.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["phase_ms"], d["roofline"]["step_us"], d["config"].get("persist"))' 2>&1 | tail -n 1)"; done
