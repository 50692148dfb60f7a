# round 4 collection: whole GPU suite, smoke, default bench line, rocprofv3 kernel stats (B = 1 / 64 / C5 share), PMC passes (HBM
# traffic of the AR step at B = 1; MFMA utilisation and traffic at B = 64; attention instruction mix), in-kernel timelines.
#   gpurun --timeout 2700 -- 'bash tools/gpu_r4_collect.sh <tag> [quick]'      (judged copies go to profiles/r04_*)
cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
QUICK=$2
export TMPDIR=/tmp
if [ -z "$QUICK" ]; then
  timeout 1500 python -m pytest tests -m gpu -q > $D/tests_all.log 2>&1; echo "all tests rc=$?"; tail -n 3 $D/tests_all.log
  python -c "import __graft_entry__ as g; g.smoke()" > $D/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $D/smoke.log
fi
timeout 900 python bench.py > $D/bench_default.log 2> $D/bench_default.err; echo "default bench rc=$?"; tail -n 1 $D/bench_default.log | cut -c1-400
timeout 300 python bench.py --steps 5 --warmup 2 --cpu-frames 0 --no-side --dtype fp8w > $D/bench_b1_fp8w.log 2>/dev/null; tail -n 1 $D/bench_b1_fp8w.log | cut -c1-200
timeout 300 python bench.py --batch 8 --steps 3 --warmup 1 --cpu-frames 0 --no-side > $D/bench_b8.log 2>/dev/null; tail -n 1 $D/bench_b8.log | cut -c1-200
timeout 300 python bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-side --opt persist=0 > $D/bench_b1_chain.log 2>/dev/null; tail -n 1 $D/bench_b1_chain.log | cut -c1-200
# kernel stats (rocprofv3 serialises the launches of a graph replay: per-kernel durations only)
(cd /tmp && rm -rf /tmp/prof1 && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof1 -o b1 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-side > $GRAFT_REPO_ROOT/$D/prof1.log 2>&1); echo "prof1 rc=$?"
cp /tmp/prof1/b1_kernel_stats.csv $D/ 2>/dev/null
(cd /tmp && rm -rf /tmp/prof64 && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof64 -o b64 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --batch 64 --steps 1 --warmup 0 --cpu-frames 0 --no-side > $GRAFT_REPO_ROOT/$D/prof64.log 2>&1); echo "prof64 rc=$?"
cp /tmp/prof64/b64_kernel_stats.csv $D/ 2>/dev/null
(cd /tmp && rm -rf /tmp/profc5 && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/profc5 -o c5 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --d-model 1536 --layers 24 --nhead 16 --dtype fp8 --batch 32 --steps 1 --warmup 0 --cpu-frames 0 --no-side > $GRAFT_REPO_ROOT/$D/profc5.log 2>&1); echo "profc5 rc=$?"
cp /tmp/profc5/c5_kernel_stats.csv $D/ 2>/dev/null
# PMC: one counter set per run, --kernel-trace only
for SET in "FETCH_SIZE" "WRITE_SIZE"; do
  (cd /tmp && rm -rf /tmp/pmc_run && timeout 400 rocprofv3 --pmc $SET --kernel-trace -d /tmp/pmc_run -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --cpu-frames 0 --no-graph --no-side > $GRAFT_REPO_ROOT/$D/pmc_$SET.log 2>&1); echo "pmc $SET rc=$?"
  python tools/pmc_summary.py /tmp/pmc_run/p_counter_collection.csv $D/pmc_${SET}_by_kernel.csv
done
for SET in "FETCH_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
  TAG=$(echo $SET | tr ' ' '+')
  (cd /tmp && rm -rf /tmp/pmc_run && timeout 500 rocprofv3 --pmc $SET --kernel-trace -d /tmp/pmc_run -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --batch 64 --steps 1 --warmup 0 --cpu-frames 0 --no-graph --no-side > $GRAFT_REPO_ROOT/$D/pmc_b64_$TAG.log 2>&1); echo "pmc b64 $TAG rc=$?"
  python tools/pmc_summary.py /tmp/pmc_run/p_counter_collection.csv $D/pmc_b64_${TAG}_by_kernel.csv
done
timeout 600 python tools/persist_probe.py --out $D --steps 300 --rounds 2 --check-steps 64 --variants pf=3 pf=3,sample=0 pf=3,steps=8 pf=3,mode=0x114 pf=0 --trace pf=3 > $D/persist_probe.log 2>&1; echo "persist probe rc=$?"
timeout 200 python tools/persist_stress.py 60 > $D/persist_stress.log 2>&1; echo "persist stress rc=$?"; tail -n 1 $D/persist_stress.log
timeout 600 python tools/ktrace_step.py --out $D/ktrace_b64 --spg 8 --batch 64 > $D/ktrace_b64.log 2>&1; echo "ktrace b64 rc=$?"
