cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "fused_qkv" > $D/t_ops.log 2>&1; echo "ops rc=$?"; tail -n 3 $D/t_ops.log
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_parity_sizes_gpu.py -x -q -k "c2 or c5 or reserve or nar_force or graph" > $D/t_eng.log 2>&1; echo "engine rc=$?"; tail -n 5 $D/t_eng.log; grep "forced batch" $D/t_eng.log
timeout 400 python tools/ar_tune.py --steps 300 --rounds 2 --variants qkv_attn=0 qa_nsplit=4 qa_waves=4 qa_waves=4,qa_nsplit=4 qa_nsplit=16 > $D/ar_tune.log 2>&1; echo "ar_tune rc=$?"; tail -n 1 $D/ar_tune.log
timeout 400 python tools/ar_tune.py --steps 740 --rounds 2 --variants qkv_attn=0 qa_nsplit=4 qa_waves=4,qa_nsplit=4 > $D/ar_tune740.log 2>&1; echo "ar_tune rc=$?"; tail -n 1 $D/ar_tune740.log
timeout 200 python tools/ar_tune.py --steps 300 --rounds 2 --dtype fp32 --variants qkv_attn=0 qa_nsplit=4 qa_waves=4 > $D/ar_tune_fp32.log 2>&1; tail -n 1 $D/ar_tune_fp32.log
timeout 200 python tools/ar_tune.py --steps 300 --rounds 2 --dtype fp8w --variants qkv_attn=0 qa_nsplit=4 > $D/ar_tune_fp8w.log 2>&1; tail -n 1 $D/ar_tune_fp8w.log
timeout 300 python tools/ktrace_step.py --out $D/ktrace_b1 --spg 8 --opt qa_nsplit=4 > $D/ktrace.log 2>&1; echo "ktrace rc=$?"; tail -c 1500 $D/ktrace.log
