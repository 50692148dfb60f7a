cd $GRAFT_REPO_ROOT
D=gpurun_out/r2s; mkdir -p $D
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_fp8_gpu.py -q -k "gemm or linear or fp8" > $D/tests_gemm.log 2>&1; echo "gemm tests rc=$?"; tail -n 5 $D/tests_gemm.log
timeout 400 python tools/gemm_bench.py > $D/gemm_bench_epi2.log 2>&1; cat $D/gemm_bench_epi2.log
timeout 600 python bench.py --batch 64 --steps 2 --warmup 1 --cpu-frames 0 > $D/bench_b64.log 2>&1; grep -o '"value": [0-9.]*\|"phase_ms": {[^}]*}' $D/bench_b64.log | head -2
timeout 600 python bench.py --batch 64 --steps 2 --warmup 1 --cpu-frames 0 --opt g8_colgroup=4 > $D/bench_b64_cg4.log 2>&1; grep -o '"value": [0-9.]*\|"phase_ms": {[^}]*}' $D/bench_b64_cg4.log | head -2
timeout 900 python bench.py --steps 2 --warmup 1 --cpu-frames 0 --d-model 1536 --layers 24 --nhead 16 --dtype fp8 --batch 32 > $D/bench_c5_fp8.log 2>&1; grep -o '"value": [0-9.]*\|"phase_ms": {[^}]*}' $D/bench_c5_fp8.log | head -2
timeout 300 python bench.py --steps 5 --warmup 2 --cpu-frames 0 --no-c3 > $D/bench_b1.log 2>&1; grep -o '"value": [0-9.]*\|"phase_ms": {[^}]*}' $D/bench_b1.log | head -2
