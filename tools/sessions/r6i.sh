# round 6, session i: batch 3 / 4 / 8 on the GEMV (skinny.hip) chain vs the MFMA skinny chain; slot-mode assertion; smoke
cd $GRAFT_REPO_ROOT
D=gpurun_out/r6i; mkdir -p $D
export TMPDIR=/tmp
timeout 300 python tools/fresh_box_probe.py --out $D/first > $D/probe.log 2>&1; echo "probe rc=$?"
timeout 600 python -m pytest tests/test_persist_gpu.py -x -q -k "default_where_covered or two_utterances" > $D/tests.log 2>&1; echo "tests rc=$?"; tail -n 2 $D/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $D/smoke.log 2>&1; echo "smoke rc=$?"
for b in 3 4 8; do for o in 0 1; do
  timeout 300 python bench.py --batch $b --no-side --cpu-frames 0 --steps 3 --warmup 1 --opt no_gemm_skinny=$o > $D/bench_b${b}_$o.log 2>&1
  echo "batch $b no_gemm_skinny=$o: $(tail -n 1 $D/bench_b${b}_$o.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["phase_ms"], d["roofline"]["step_us"])' 2>&1 | tail -n 1)"
done; done
