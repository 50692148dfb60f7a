cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
for o in "gs_wfrag_test=0" "gs_wfrag_test=1" "gs_wfrag_test=0" "gs_wfrag_test=1"; do
  timeout 300 python bench.py --batch 64 --steps 1 --warmup 1 --cpu-frames 0 --opt $o > $D/bench_b64_$o.log 2>&1; echo "b64 $o rc=$?"; tail -n 1 $D/bench_b64_$o.log | grep -o '"phase_ms[^}]*}'
done
