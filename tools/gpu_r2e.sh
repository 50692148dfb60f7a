# round 2, call E: FP8 mode (fp8 MFMA GEMM) parity + C5-share timing, then the whole GPU suite
cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fp8_gpu.py -q -x -s > $D/tests_fp8.log 2>&1; echo "fp8 tests rc=$?"; grep -E "FP8 engine|passed|failed|Error|assert" $D/tests_fp8.log | tail -n 12
for DT in fp8 fp8w; do timeout 600 python bench.py --steps 2 --warmup 1 --cpu-frames 0 --d-model 1536 --layers 24 --nhead 16 --dtype $DT --batch 32 > $D/bench_c5_$DT.log 2>&1; echo "c5 $DT $(tail -n 1 $D/bench_c5_$DT.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["phase_ms"], d["roofline"]["launch_us"], d["roofline"]["frac"])')"; done
timeout 1200 python -m pytest tests -m gpu -q > $D/tests_all.log 2>&1; echo "all tests rc=$?"; tail -n 6 $D/tests_all.log
