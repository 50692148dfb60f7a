# round 3, second collection (after the compile-time-layout bodies of the batched GEMMs; the batch-1 kernels -- and with them
# profiles/r03_b1_*, r03_pmc_*, r03_ktrace_b1_* and ar_step_traffic.json -- are unchanged): whole GPU suite, smoke, default bench
# line, B = 8 line, rocprofv3 kernel stats at B = 64, in-kernel timeline + per-wave distribution at B = 64, the load-burst ubench.
#   gpurun -- 'bash tools/gpu_r3_collect2.sh <tag>'   (judged copies go to profiles/r03_*)
cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $D/tests_all.log 2>&1; echo "all tests rc=$?"; tail -n 3 $D/tests_all.log
python -c "import __graft_entry__ as g; g.smoke()" > $D/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $D/smoke.log
timeout 900 python bench.py > $D/bench_default.log 2>&1; echo "default bench rc=$?"; tail -n 1 $D/bench_default.log | cut -c1-600
timeout 300 python bench.py --batch 8 --steps 3 --warmup 1 --cpu-frames 0 --no-side > $D/bench_b8.log 2>&1; tail -n 1 $D/bench_b8.log | cut -c1-200
(cd /tmp && rm -rf /tmp/prof64 && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof64 -o b64 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --batch 64 --steps 1 --warmup 0 --cpu-frames 0 --no-side > $GRAFT_REPO_ROOT/$D/prof64.log 2>&1); echo "prof64 rc=$?"
cp /tmp/prof64/b64_kernel_stats.csv $D/ 2>/dev/null
timeout 600 python tools/ktrace_step.py --out $D/ktrace_b64 --spg 8 --batch 64 > $D/ktrace_b64.log 2>&1; echo "ktrace b64 rc=$?"
timeout 300 python tools/ktrace_dist.py --out $D/ktrace_dist_b64.json > $D/ktrace_dist.log 2>&1; echo "ktrace dist rc=$?"
timeout 200 tools/bin/ubench_xload > $D/ubench_xload.json 2> $D/ubench_xload.err; echo "ubench xload rc=$?"
