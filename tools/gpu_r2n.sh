cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
timeout 900 python -m pytest tests/test_serving_gpu.py tests/test_engine_gpu.py tests/test_ops_gpu.py -q -k "serving or slot or sampl or grows or handoff or continuous or reproducible" > $D/tests.log 2>&1; echo "tests rc=$?"; tail -n 25 $D/tests.log
