cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
timeout 900 python -m pytest tests -m gpu -x -q > $D/tests.log 2>&1; echo "tests rc=$?"; tail -n 6 $D/tests.log
timeout 400 python bench.py --steps 5 --warmup 2 > $D/bench.log 2>&1; echo "bench rc=$?"; tail -n 1 $D/bench.log
