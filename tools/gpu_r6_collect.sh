# round 6 collection: instrumented first process, whole GPU suite, smoke, default bench line, the side benches, rocprofv3 kernel stats
# (B = 1 / 64), PMC passes (HBM traffic of the AR step at B = 1; MFMA utilisation at B = 64), persistent-step probe (bit-identity ladder,
# timing, timeline) and stress.   gpurun --timeout 2700 -- 'bash tools/gpu_r6_collect.sh <tag>'   (judged copies go to profiles/r06_*)
cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
( timeout 300 python tools/fresh_box_probe.py --out $D/first > $D/first.out 2> $D/first.err ); echo "first-process probe rc=$?"
timeout 1500 python -m pytest tests -m gpu -q > $D/tests_all.log 2>&1; echo "all tests rc=$?"; tail -n 3 $D/tests_all.log
python -c "import __graft_entry__ as g; g.smoke()" > $D/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 3 $D/smoke.log
timeout 900 python bench.py > $D/bench_default.log 2> $D/bench_default.err; echo "default bench rc=$?"; tail -n 1 $D/bench_default.log | cut -c1-300
timeout 300 python bench.py --steps 5 --warmup 2 --cpu-frames 0 --no-side --dtype fp8w > $D/bench_b1_fp8w.log 2>/dev/null; tail -n 1 $D/bench_b1_fp8w.log | cut -c1-200
timeout 300 python bench.py --batch 8 --steps 3 --warmup 1 --cpu-frames 0 --no-side > $D/bench_b8.log 2>/dev/null; tail -n 1 $D/bench_b8.log | cut -c1-200
for b in 2 3 4 5 6; do timeout 300 python bench.py --batch $b --steps 3 --warmup 1 --cpu-frames 0 --no-side > $D/bench_b$b.log 2>/dev/null; tail -n 1 $D/bench_b$b.log | cut -c1-200; done
timeout 300 python bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-side --dtype fp32 > $D/bench_b1_fp32.log 2>/dev/null; tail -n 1 $D/bench_b1_fp32.log | cut -c1-200
timeout 300 python bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-side --opt persist=0 > $D/bench_b1_chain.log 2>/dev/null; tail -n 1 $D/bench_b1_chain.log | cut -c1-200
(cd /tmp && rm -rf /tmp/prof1 && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof1 -o b1 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-side > $GRAFT_REPO_ROOT/$D/prof1.log 2>&1); echo "prof1 rc=$?"
cp $(find /tmp/prof1 -name "*kernel_stats.csv" | head -1) $D/b1_kernel_stats.csv 2>/dev/null
(cd /tmp && rm -rf /tmp/prof64 && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof64 -o b64 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --batch 64 --steps 1 --warmup 0 --cpu-frames 0 --no-side > $GRAFT_REPO_ROOT/$D/prof64.log 2>&1); echo "prof64 rc=$?"
cp $(find /tmp/prof64 -name "*kernel_stats.csv" | head -1) $D/b64_kernel_stats.csv 2>/dev/null
for SET in "FETCH_SIZE" "WRITE_SIZE"; do
  (cd /tmp && rm -rf /tmp/pmc_run && timeout 400 rocprofv3 --pmc $SET --kernel-trace -d /tmp/pmc_run -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --cpu-frames 0 --no-graph --no-side > $GRAFT_REPO_ROOT/$D/pmc_$SET.log 2>&1); echo "pmc $SET rc=$?"
  python tools/pmc_summary.py $(find /tmp/pmc_run -name "*counter_collection.csv" | head -1) $D/pmc_${SET}_by_kernel.csv
done
SET="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"
(cd /tmp && rm -rf /tmp/pmc_run && timeout 500 rocprofv3 --pmc $SET --kernel-trace -d /tmp/pmc_run -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --batch 64 --steps 1 --warmup 0 --cpu-frames 0 --no-graph --no-side > $GRAFT_REPO_ROOT/$D/pmc_b64_MFMA_BUSY.log 2>&1); echo "pmc b64 MFMA_BUSY rc=$?"
python tools/pmc_summary.py $(find /tmp/pmc_run -name "*counter_collection.csv" | head -1) $D/pmc_b64_MFMA_BUSY_by_kernel.csv
timeout 700 python tools/persist_probe.py --out $D --steps 400 --rounds 2 --check-steps 64 --variants pf=3 pf=3,mode=0x134 pf=3,sample=0 pf=3,steps=8 pf=3,mode=0x114 pf=0 pf=3,naps=0x335856 pf=3,naps=0x325856 pf=3,naps=0x325757 --trace pf=3 > $D/persist_probe.log 2>&1; echo "persist probe rc=$?"; grep "\[time\]" $D/persist_probe.log | tail -1 | cut -c1-600
timeout 200 python tools/persist_stress.py 40 > $D/persist_stress.log 2>&1; echo "persist stress rc=$?"; tail -n 1 $D/persist_stress.log
# the batched persistent launch (persist_nb.hip): kernel stats and HBM traffic at 4 utterances, step time / timeline at 2 .. 4
(cd /tmp && rm -rf /tmp/prof4 && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof4 -o b4 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --batch 4 --steps 2 --warmup 1 --cpu-frames 0 --no-side > $GRAFT_REPO_ROOT/$D/prof4.log 2>&1); echo "prof4 rc=$?"
cp $(find /tmp/prof4 -name "*kernel_stats.csv" | head -1) $D/b4_kernel_stats.csv 2>/dev/null
for SET in "FETCH_SIZE" "WRITE_SIZE"; do
  (cd /tmp && rm -rf /tmp/pmc_run && timeout 400 rocprofv3 --pmc $SET --kernel-trace -d /tmp/pmc_run -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --batch 4 --steps 1 --warmup 1 --cpu-frames 0 --no-graph --no-side > $GRAFT_REPO_ROOT/$D/pmc_b4_$SET.log 2>&1); echo "pmc b4 $SET rc=$?"
  python tools/pmc_summary.py $(find /tmp/pmc_run -name "*counter_collection.csv" | head -1) $D/pmc_b4_${SET}_by_kernel.csv
done
timeout 600 python tools/persist_nb_probe.py --out $D --trace > $D/persist_nb_probe.log 2>&1; echo "persist nb probe rc=$?"; grep "\[time\]" $D/persist_nb_probe.log | cut -c1-300
timeout 300 python tools/persist_nb_stress.py 150 > $D/persist_nb_stress.log 2>&1; echo "persist nb stress rc=$?"; tail -n 1 $D/persist_nb_stress.log
