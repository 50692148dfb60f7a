#!/bin/bash
# round 3: FAST body of gemm_skinny -- parity, then the B = 64 timeline and bench
D=gpurun_out/r3v; mkdir -p $D
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_ops_gpu.py -q -k "compile_time_layout or fused_layernorm or fused_out_proj or batch_path or ragged_batch or m_split or split_k" 2>&1 | tail -8
timeout 300 python tools/ktrace_dist.py --out $D/ktrace_dist.json > $D/ktrace_dist.log 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3v/ktrace_dist.json'))
for k,v in d.items():
    print(k, v['waves_per_launch'], 'mark1', v['mark1_after_own_start_us'], 'end', v['end_after_own_start_us'])
PY
timeout 300 python tools/ktrace_step.py --out $D/ktrace_b64 --spg 8 --batch 64 > $D/ktrace_b64.log 2>&1; echo "ktrace b64 rc=$?"
python - <<'PY'
import json
s=json.load(open('gpurun_out/r3v/ktrace_b64_summary.json'))
print(s['ar_us_per_step_default'])
for k,v in s['families'].items(): print(k, v['mean_gap_us'], v['mean_body_us'])
PY
timeout 400 python bench.py --batch 64 --steps 2 --warmup 1 --cpu-frames 0 --no-side > $D/bench_b64.log 2>&1; tail -n 1 $D/bench_b64.log | cut -c1-200
