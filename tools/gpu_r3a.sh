# Round 3, call A: new kernels' unit tests + engine parity at C2, A/B of the fused QKV + attention launch, boundary bisect.
cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "fused_qkv or split_k_handoff or decode_attention" > $D/t_ops.log 2>&1; echo "ops rc=$?"; tail -n 3 $D/t_ops.log
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_parity_sizes_gpu.py tests/test_options_gpu.py -x -q -k "c2 or c3 or c5 or seed or reserve or nar_force or graph or golden" > $D/t_eng.log 2>&1; echo "engine rc=$?"; tail -n 5 $D/t_eng.log
timeout 300 python tools/ar_tune.py --steps 300 --rounds 3 > $D/ar_tune.log 2>&1; echo "ar_tune rc=$?"; tail -n 1 $D/ar_tune.log
timeout 200 python tools/ar_tune.py --steps 300 --rounds 2 --dtype fp32 --variants qkv_attn=0 > $D/ar_tune_fp32.log 2>&1; tail -n 1 $D/ar_tune_fp32.log
timeout 200 python tools/ar_tune.py --steps 300 --rounds 2 --dtype fp8w --variants qkv_attn=0 > $D/ar_tune_fp8w.log 2>&1; tail -n 1 $D/ar_tune_fp8w.log
for env in "X=1" "HIP_FORCE_DEV_KERNARG=0" "HIP_FORCE_DEV_KERNARG=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1"; do
  env $env timeout 120 tools/bin/ubench_boundary "$env" > $D/ub_$env.json 2>&1; echo "ub $env rc=$?"
done
timeout 120 tools/bin/ubench_boundary_preload preload > $D/ub_preload.json 2>&1; echo "ub preload rc=$?"
for env in "HIP_FORCE_DEV_KERNARG=0" "HIP_FORCE_DEV_KERNARG=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"; do
  env $env timeout 200 python tools/ar_tune.py --steps 300 --rounds 2 --variants qkv_attn=0 > $D/ar_tune_$env.log 2>&1; echo "$env: $(tail -n 1 $D/ar_tune_$env.log)"
done
timeout 300 python tools/ktrace_step.py --out $D/ktrace_b1 --spg 8 > $D/ktrace.log 2>&1; echo "ktrace rc=$?"; tail -c 1500 $D/ktrace.log
