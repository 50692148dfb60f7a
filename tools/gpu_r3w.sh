#!/bin/bash
# round 3: FAST bodies incl. d = 1536 -- parity, C5 share and B = 8 benches
D=gpurun_out/r3w; mkdir -p $D
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_ops_gpu.py -q -k "compile_time_layout or fused_layernorm or fused_out_proj or batch_path or ragged_batch or m_split or split_k" 2>&1 | tail -8
timeout 300 python bench.py --batch 8 --steps 3 --warmup 1 --cpu-frames 0 --no-side > $D/bench_b8.log 2>&1; tail -n 1 $D/bench_b8.log | cut -c1-200
timeout 600 python bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-fp32 > $D/bench_side.log 2>&1; tail -n 1 $D/bench_side.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'])
for k in ('c3_batch64','c5_share_fp8'):
    print(k, d[k]['value'], d[k].get('phase_ms'))
"
