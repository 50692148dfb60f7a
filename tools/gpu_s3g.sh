cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fp8w_gpu.py -m gpu -q > $D/tests_fp8.log 2>&1; echo "fp8w tests rc=$?"; tail -n 6 $D/tests_fp8.log | cut -c1-300
timeout 900 python -m pytest tests -m gpu -x -q > $D/tests.log 2>&1; echo "tests rc=$?"; tail -n 4 $D/tests.log
# BASELINE configs[4] per-GPU share: d1536-L24-h16, fp8 weights, 32 utterances
timeout 900 python bench.py --d-model 1536 --layers 24 --nhead 16 --dtype fp8w --batch 32 --steps 1 --warmup 1 --cpu-frames 0 > $D/bench_c5_b32.log 2>&1; echo "c5 b32 rc=$?"; tail -n 1 $D/bench_c5_b32.log | cut -c1-1500
timeout 900 python bench.py --d-model 1536 --layers 24 --nhead 16 --dtype bf16 --batch 32 --steps 1 --warmup 1 --cpu-frames 0 > $D/bench_c5_b32_bf16.log 2>&1; echo "c5 b32 bf16 rc=$?"; tail -n 1 $D/bench_c5_b32_bf16.log | grep -o '"value": [0-9.]*\|"phase_ms[^}]*}'
timeout 900 python bench.py --d-model 1536 --layers 24 --nhead 16 --dtype fp8w --batch 1 --steps 2 --warmup 1 --cpu-frames 0 --no-c3 > $D/bench_c5_b1.log 2>&1; echo "c5 b1 rc=$?"; tail -n 1 $D/bench_c5_b1.log | grep -o '"value": [0-9.]*\|"phase_ms[^}]*}'
