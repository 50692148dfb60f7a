cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "m_split or split_k or linear" > $D/t_ops.log 2>&1; echo "ops rc=$?"; tail -n 3 $D/t_ops.log
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_parity_sizes_gpu.py tests/test_serving_gpu.py tests/test_fp8w_gpu.py -x -q > $D/t_eng.log 2>&1; echo "engine rc=$?"; tail -n 3 $D/t_eng.log
for opt in "" "--opt gs_msplit=0"; do
  timeout 300 python bench.py --batch 64 --steps 2 --warmup 1 --cpu-frames 0 --no-c3 --no-fp32 $opt > $D/b64_$(echo $opt | tr -d ' =-').log 2>&1; tail -n 1 $D/b64_$(echo $opt | tr -d ' =-').log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$opt', d['value'], d['phase_ms'], d['roofline']['launch_us'])"
done
for opt in "" "--opt gs_msplit=0"; do
  timeout 300 python bench.py --batch 8 --steps 3 --warmup 1 --cpu-frames 0 --no-c3 --no-fp32 $opt > $D/b8.log 2>&1; tail -n 1 $D/b8.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('b8 $opt', d['value'], d['phase_ms'], d['roofline']['launch_us'])"
done
timeout 600 python tools/ktrace_step.py --out $D/ktrace_b64 --spg 8 --batch 64 > $D/ktrace_b64.log 2>&1; echo "ktrace b64 rc=$?"; head -7 $D/ktrace_b64_timeline.csv
