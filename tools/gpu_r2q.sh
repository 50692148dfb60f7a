cd $GRAFT_REPO_ROOT
D=gpurun_out/r2q; mkdir -p $D
timeout 900 python -m pytest tests/test_ops_gpu.py -q -k "gemm_tile_policy or linear" > $D/tests_gemm.log 2>&1; echo "gemm tests rc=$?"; tail -n 5 $D/tests_gemm.log
timeout 400 python tools/gemm_bench.py > $D/gemm_bench_epi.log 2>&1; cat $D/gemm_bench_epi.log
