cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_engine_gpu.py -x -q -k "handoff_agree" > $D/t.log 2>&1; echo "test rc=$?"; tail -n 2 $D/t.log
timeout 300 python tools/ar_tune.py --steps 740 --rounds 2 --variants qa_nsplit=4,qa_nk=8 qa_nk=8 qa_nsplit=4 > $D/ar_tune740.log 2>&1; tail -n 1 $D/ar_tune740.log
timeout 300 python tools/ar_tune.py --steps 300 --rounds 2 --variants qa_nsplit=4,qa_nk=8 qa_nk=8 qa_nsplit=4 > $D/ar_tune300.log 2>&1; tail -n 1 $D/ar_tune300.log
