# round 6, session g: fp32 packed-row GEMMs on the LDS-DMA ring -- bit identity, the token-exact parity tests, fp32 bench A/B
cd $GRAFT_REPO_ROOT
D=gpurun_out/r6g; mkdir -p $D
export TMPDIR=/tmp
timeout 300 python tools/fresh_box_probe.py --out $D/first > $D/probe.log 2>&1; echo "probe rc=$?"
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "fp32_gemm_on_the_lds or linear" > $D/tests_ops.log 2>&1; echo "ops tests rc=$?"; tail -n 3 $D/tests_ops.log
timeout 900 python -m pytest tests/test_persist_gpu.py -x -q -k "two_utterances" > $D/tests_p.log 2>&1; echo "b2 test rc=$?"; tail -n 2 $D/tests_p.log
for v in 0 1 0 1; do
  timeout 300 python bench.py --dtype fp32 --no-side --cpu-frames 0 --steps 4 --warmup 1 --opt f32_glds=$v > $D/bench_fp32_$v.log 2>&1
  echo "f32_glds=$v: $(tail -n 1 $D/bench_fp32_$v.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["phase_ms"])' 2>&1 | tail -n 1)"
done
timeout 1800 python -m pytest tests/test_parity_sizes_gpu.py tests/test_engine_gpu.py tests/test_eos_gpu.py tests/test_options_gpu.py tests/test_modules_gpu.py tests/test_forward_gpu.py -x -q > $D/tests_parity.log 2>&1; echo "parity tests rc=$?"; tail -n 3 $D/tests_parity.log
