cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "tile_policy" > $D/tests_gemm.log 2>&1; echo "gemm tests rc=$?"; tail -n 3 $D/tests_gemm.log | cut -c1-300
timeout 300 python tools/gemm_bench.py > $D/gemm_bench.log 2>&1; echo "gemm bench rc=$?"; grep -v amdgpu $D/gemm_bench.log
