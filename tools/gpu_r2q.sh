cd $GRAFT_REPO_ROOT
D=gpurun_out/r2q; mkdir -p $D
timeout 600 python -m pytest tests/test_ops_gpu.py -q -k attention > $D/tests_attn.log 2>&1; echo "attn tests rc=$?"; tail -n 5 $D/tests_attn.log
timeout 400 python tools/attn_bench.py > $D/attn_bench2.log 2>&1; cat $D/attn_bench2.log
