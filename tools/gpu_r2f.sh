# round 2, call F: FP8 tests again, batched-step in-kernel timeline (C3 and C5-shaped), rocprof stats of the FP8 C5 share
cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_fp8_gpu.py -q -x -s > $D/tests_fp8.log 2>&1; echo "fp8 tests rc=$?"; grep -E "FP8 engine|passed|failed|Error|assert" $D/tests_fp8.log | tail -n 8
timeout 600 python tools/ktrace_step.py --out $D/ktrace_b64 --spg 8 --batch 64 > $D/ktrace_b64.log 2>&1; echo "ktrace b64 rc=$?"; tail -n 1 $D/ktrace_b64.log | cut -c1-1800
timeout 600 python tools/ktrace_step.py --out $D/ktrace_c5 --spg 8 --batch 32 --d-model 1536 --layers 12 --dtype fp8w > $D/ktrace_c5.log 2>&1; echo "ktrace c5 rc=$?"; tail -n 1 $D/ktrace_c5.log | cut -c1-1800
(cd /tmp && rm -rf /tmp/pf && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pf -o c5 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --d-model 1536 --layers 24 --nhead 16 --dtype fp8 --batch 32 --steps 1 --warmup 0 --cpu-frames 0 > $GRAFT_REPO_ROOT/$D/prof_c5.log 2>&1); echo "prof rc=$?"
cp /tmp/pf/c5_kernel_stats.csv $D/c5_fp8_kernel_stats.csv 2>/dev/null; head -n 16 $D/c5_fp8_kernel_stats.csv | cut -c1-200
