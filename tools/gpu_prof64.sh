cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
(cd /tmp && rm -rf /tmp/prof64 && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof64 -o b64 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --batch 64 --steps 1 --warmup 0 --cpu-frames 0 > $GRAFT_REPO_ROOT/$D/prof64.log 2>&1); echo "prof rc=$?"
cp /tmp/prof64/b64_kernel_stats.csv $D/ 2>/dev/null
head -n 22 $D/b64_kernel_stats.csv | cut -c1-100,180-300
