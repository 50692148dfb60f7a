# round 3 final collection: whole GPU suite, smoke, default bench line, rocprofv3 kernel stats (B = 1 / 64), PMC traffic of the AR step,
# in-kernel timelines.   gpurun -- 'bash tools/gpu_r3_collect.sh <tag>'   (judged copies go to profiles/r03_*)
cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $D/tests_all.log 2>&1; echo "all tests rc=$?"; tail -n 3 $D/tests_all.log
python -c "import __graft_entry__ as g; g.smoke()" > $D/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $D/smoke.log
timeout 900 python bench.py > $D/bench_default.log 2>&1; echo "default bench rc=$?"; tail -n 1 $D/bench_default.log | cut -c1-600
timeout 300 python bench.py --steps 5 --warmup 2 --cpu-frames 0 --no-side --dtype fp8w > $D/bench_b1_fp8w.log 2>&1; tail -n 1 $D/bench_b1_fp8w.log | cut -c1-200
timeout 300 python bench.py --batch 8 --steps 3 --warmup 1 --cpu-frames 0 --no-side > $D/bench_b8.log 2>&1; tail -n 1 $D/bench_b8.log | cut -c1-200
(cd /tmp && rm -rf /tmp/prof1 && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof1 -o b1 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-side > $GRAFT_REPO_ROOT/$D/prof1.log 2>&1); echo "prof1 rc=$?"
cp /tmp/prof1/b1_kernel_stats.csv $D/ 2>/dev/null
(cd /tmp && rm -rf /tmp/prof64 && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof64 -o b64 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --batch 64 --steps 1 --warmup 0 --cpu-frames 0 --no-side > $GRAFT_REPO_ROOT/$D/prof64.log 2>&1); echo "prof64 rc=$?"
cp /tmp/prof64/b64_kernel_stats.csv $D/ 2>/dev/null
for SET in "FETCH_SIZE" "WRITE_SIZE"; do
  (cd /tmp && rm -rf /tmp/pmc_run && timeout 400 rocprofv3 --pmc $SET --kernel-trace -d /tmp/pmc_run -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --cpu-frames 0 --no-graph --no-side > $GRAFT_REPO_ROOT/$D/pmc_$SET.log 2>&1); echo "pmc $SET rc=$?"
  python tools/pmc_summary.py /tmp/pmc_run/p_counter_collection.csv $D/pmc_${SET}_by_kernel.csv
done
timeout 600 python tools/ktrace_step.py --out $D/ktrace_b1 > $D/ktrace_b1.log 2>&1; echo "ktrace b1 rc=$?"
timeout 600 python tools/ktrace_step.py --out $D/ktrace_b64 --spg 8 --batch 64 > $D/ktrace_b64.log 2>&1; echo "ktrace b64 rc=$?"
