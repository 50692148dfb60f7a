# round 6, session h: fp32 attention staging / block size
cd $GRAFT_REPO_ROOT
D=gpurun_out/r6h; mkdir -p $D
export TMPDIR=/tmp
timeout 300 python tools/fresh_box_probe.py --out $D/first > $D/probe.log 2>&1; echo "probe rc=$?"
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "attention" > $D/tests_ops.log 2>&1; echo "ops tests rc=$?"; tail -n 3 $D/tests_ops.log
for rep in 1 2; do for v in "attn_f32_vec=0 --opt attn_f32_qb=64" "attn_f32_vec=1 --opt attn_f32_qb=64" "attn_f32_vec=0 --opt attn_f32_qb=32" "attn_f32_vec=1 --opt attn_f32_qb=32"; do
  timeout 300 python bench.py --dtype fp32 --no-side --cpu-frames 0 --steps 4 --warmup 1 --opt $v > $D/bench_fp32.log 2>&1
  echo "$v: $(tail -n 1 $D/bench_fp32.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["phase_ms"])' 2>&1 | tail -n 1)"
done; done
