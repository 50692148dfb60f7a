cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
timeout 300 python tools/ar_tune.py --steps 300 --rounds 2 --variants qkv_attn=0 qa_qtemporal=0 > $D/ar_tune.log 2>&1; echo "ar_tune rc=$?"; tail -n 1 $D/ar_tune.log
timeout 300 python tools/ktrace_step.py --out $D/ktrace_b1 --spg 8 > $D/ktrace.log 2>&1; echo "ktrace rc=$?"; head -3 $D/ktrace_b1_timeline.csv
timeout 1500 python -m pytest tests -m gpu -x -q > $D/tests.log 2>&1; echo "tests rc=$?"; tail -n 4 $D/tests.log
