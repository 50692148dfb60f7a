# round 3: rocprofv3 kernel stats of the configs[4] share (d1536-L24, fp8 weights + fp8 MFMA prefill / NAR, 32 utterances)
cd $GRAFT_REPO_ROOT
D=gpurun_out/r3x; mkdir -p $D
export TMPDIR=/tmp
(cd /tmp && rm -rf /tmp/profc5 && timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/profc5 -o c5 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --batch 32 --d-model 1536 --layers 24 --dtype fp8 --steps 1 --warmup 0 --cpu-frames 0 --no-side > $GRAFT_REPO_ROOT/$D/profc5.log 2>&1); echo "profc5 rc=$?"
cp /tmp/profc5/c5_kernel_stats.csv $D/ 2>/dev/null
tail -n 1 $D/profc5.log | cut -c1-300
head -14 $D/c5_kernel_stats.csv | cut -c1-200
