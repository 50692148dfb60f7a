cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_parity_sizes_gpu.py tests/test_serving_gpu.py -x -q -k "c2 or golden or graph or serving or slot or sampled or stop" > $D/t_eng.log 2>&1; echo "engine rc=$?"; tail -n 3 $D/t_eng.log
timeout 300 python tools/ar_tune.py --steps 300 --rounds 2 --variants qa_handoff=0 qkv_attn=0 qa_nsplit=8 > $D/ar_tune.log 2>&1; echo "ar_tune rc=$?"; tail -n 1 $D/ar_tune.log
timeout 300 python tools/ar_tune.py --steps 740 --rounds 2 --variants qa_handoff=0 qa_nsplit=8 > $D/ar_tune740.log 2>&1; tail -n 1 $D/ar_tune740.log
timeout 200 python tools/ar_tune.py --steps 300 --rounds 2 --dtype fp32 --variants qa_handoff=0 qkv_attn=0 > $D/ar_tune_fp32.log 2>&1; tail -n 1 $D/ar_tune_fp32.log
timeout 200 python tools/ar_tune.py --steps 300 --rounds 2 --dtype fp8w --variants qa_handoff=0 > $D/ar_tune_fp8w.log 2>&1; tail -n 1 $D/ar_tune_fp8w.log
timeout 300 python tools/ktrace_step.py --out $D/ktrace_b1 --spg 8 > $D/ktrace.log 2>&1; head -4 $D/ktrace_b1_timeline.csv
