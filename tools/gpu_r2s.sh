cd $GRAFT_REPO_ROOT
D=gpurun_out/r2s; mkdir -p $D
timeout 900 python -m pytest tests/test_options_gpu.py -q > $D/tests_opt.log 2>&1; echo "opt tests rc=$?"; tail -n 25 $D/tests_opt.log
