# PMC passes (one counter set per rocprofv3 run, --kernel-trace only -- never with sys/hip/hsa tracing):
#   gpurun -- 'bash tools/gpu_pmc.sh <tag> [bench flags]'      default flags: the B = 1 bench, eager launches
# Per-kernel sums go to gpurun_out/<tag>/pmc_<SET>_by_kernel.csv (tools/pmc_summary.py).
cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
shift
FLAGS="${@:---steps 1 --warmup 1 --cpu-frames 0 --no-graph --no-side}"
export TMPDIR=/tmp
for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  TAG=$(echo $SET | tr ' ' '+')
  (cd /tmp && rm -rf /tmp/pmc_run && timeout 400 rocprofv3 --pmc $SET --kernel-trace -d /tmp/pmc_run -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py $FLAGS > $GRAFT_REPO_ROOT/$D/pmc_$TAG.log 2>&1); echo "pmc $TAG rc=$?"
  python tools/pmc_summary.py /tmp/pmc_run/p_counter_collection.csv $D/pmc_${TAG}_by_kernel.csv
done
