cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rm -rf /tmp/pmc_$C && timeout 300 rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc_$C -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --cpu-frames 0 --no-graph > $GRAFT_REPO_ROOT/$D/pmc_$C.log 2>&1); echo "pmc $C rc=$?"
  python tools/pmc_summary.py /tmp/pmc_$C/p_counter_collection.csv $D/pmc_${C}_by_kernel.csv
  head -n 3 /tmp/pmc_$C/p_counter_collection.csv
done
