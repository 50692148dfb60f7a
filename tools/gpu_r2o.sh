cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
for DBG in 0 1 2 3; do
timeout 600 python tools/ktrace_step.py --out $D/ktrace_b64_dbg$DBG --spg 8 --batch 64 --opt gs_dbg=$DBG > $D/ktrace_b64_dbg$DBG.log 2>&1; echo "dbg=$DBG rc=$?"; sed -n 2,6p $D/ktrace_b64_dbg${DBG}_timeline.csv
done
