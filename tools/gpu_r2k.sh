cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "attention" > $D/tests_attn.log 2>&1; echo "attn tests rc=$?"; tail -n 3 $D/tests_attn.log
timeout 300 python tools/attn_bench.py > $D/attn_bench.log 2>&1; cat $D/attn_bench.log | tail -n 5
