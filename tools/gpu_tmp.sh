cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "sampling_distribution or sampled" > $D/tests_samp.log 2>&1; echo "sampling tests rc=$?"; tail -n 8 $D/tests_samp.log | cut -c1-300
