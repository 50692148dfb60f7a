# round 2: whole GPU suite, smoke, bench lines, rocprof kernel stats, PMC traffic, in-kernel timeline, microbenchmarks.
#   gpurun -- 'bash tools/gpu_r2_collect.sh <tag>'     (results under gpurun_out/<tag>/; the judged copies go to profiles/)
cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $D/tests_all.log 2>&1; echo "all tests rc=$?"; tail -n 5 $D/tests_all.log
python -c "import __graft_entry__ as g; g.smoke()" > $D/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 3 $D/smoke.log
timeout 900 python bench.py > $D/bench_default.log 2>&1; echo "default bench rc=$?"; tail -n 1 $D/bench_default.log | cut -c1-700
timeout 300 python bench.py --steps 3 --warmup 1 --cpu-frames 0 --batch 8 > $D/bench_b8.log 2>&1; tail -n 1 $D/bench_b8.log | cut -c1-300
timeout 900 python bench.py --steps 2 --warmup 1 --cpu-frames 0 --d-model 1536 --layers 24 --nhead 16 --dtype fp8 --batch 32 > $D/bench_c5_fp8.log 2>&1; tail -n 1 $D/bench_c5_fp8.log | cut -c1-500
timeout 300 python bench.py --steps 5 --warmup 2 --cpu-frames 0 --no-c3 --dtype fp8w > $D/bench_b1_fp8w.log 2>&1; tail -n 1 $D/bench_b1_fp8w.log | cut -c1-300
(cd /tmp && rm -rf /tmp/prof1 && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof1 -o b1 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-c3 > $GRAFT_REPO_ROOT/$D/prof1.log 2>&1); echo "prof1 rc=$?"
cp /tmp/prof1/b1_kernel_stats.csv $D/ 2>/dev/null
(cd /tmp && rm -rf /tmp/prof64 && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof64 -o b64 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --batch 64 --steps 1 --warmup 0 --cpu-frames 0 > $GRAFT_REPO_ROOT/$D/prof64.log 2>&1); echo "prof64 rc=$?"
cp /tmp/prof64/b64_kernel_stats.csv $D/ 2>/dev/null
for SET in "FETCH_SIZE" "WRITE_SIZE"; do
  (cd /tmp && rm -rf /tmp/pmc_run && timeout 400 rocprofv3 --pmc $SET --kernel-trace -d /tmp/pmc_run -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --cpu-frames 0 --no-graph --no-c3 > $GRAFT_REPO_ROOT/$D/pmc_$SET.log 2>&1); echo "pmc $SET rc=$?"
  python tools/pmc_summary.py /tmp/pmc_run/p_counter_collection.csv $D/pmc_${SET}_by_kernel.csv
done
for SET in "FETCH_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  TAG=$(echo $SET | tr ' ' '+')
  (cd /tmp && rm -rf /tmp/pmc_run && timeout 400 rocprofv3 --pmc $SET --kernel-trace -d /tmp/pmc_run -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --batch 64 --steps 1 --warmup 0 --cpu-frames 0 --no-graph > $GRAFT_REPO_ROOT/$D/pmc_b64_$TAG.log 2>&1); echo "pmc b64 $TAG rc=$?"
  python tools/pmc_summary.py /tmp/pmc_run/p_counter_collection.csv $D/b64_pmc_${TAG}_by_kernel.csv
done
timeout 600 python tools/ktrace_step.py --out $D/ktrace_b1 > $D/ktrace_b1.log 2>&1; echo "ktrace b1 rc=$?"
timeout 600 python tools/ktrace_step.py --out $D/ktrace_b64 --spg 8 --batch 64 > $D/ktrace_b64.log 2>&1; echo "ktrace b64 rc=$?"
timeout 120 tools/bin/ubench_edges > $D/ubench_edges.json 2> $D/ubench_edges.err; echo "edges rc=$?"
timeout 300 python tools/attn_bench.py > $D/attn_bench.log 2>&1; grep -E "^C|default" $D/attn_bench.log
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INSTS_SALU"; do
  TAG=$(echo $SET | tr ' ' '+')
  (cd /tmp && rm -rf /tmp/pmc_run && timeout 200 rocprofv3 --pmc $SET --kernel-trace -d /tmp/pmc_run -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/attn_bench.py --quick > $GRAFT_REPO_ROOT/$D/pmc_attn_$TAG.log 2>&1); echo "pmc attn $TAG rc=$?"
  python tools/pmc_summary.py /tmp/pmc_run/p_counter_collection.csv $D/attn_pmc_${TAG}.csv
done
timeout 300 python tools/gemm_bench.py > $D/gemm_bench.log 2>&1; tail -n 12 $D/gemm_bench.log
timeout 120 tools/bin/ubench_l2keep > $D/ubench_l2keep.json 2> $D/ubench_l2keep.err; echo "l2keep rc=$?"
timeout 200 tools/bin/ubench_prefetch > $D/ubench_prefetch.json 2> $D/ubench_prefetch.err; echo "prefetch rc=$?"
