#!/bin/bash
# Round 5, GPU session O: per-mode first-sweep waits as defaults -- persist tests, the three one-utterance bench lines.
O=gpurun_out/r5o; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python tools/fresh_box_probe.py --out $O/first > $O/first.out 2> $O/first.err ) ; echo "first-process probe rc=$?" >> $O/log
( timeout 600 python -m pytest tests/test_persist_gpu.py -x -q -m gpu > $O/pytest_persist.log 2>&1 ) ; echo "pytest persist rc=$?" >> $O/log
( timeout 300 python bench.py --steps 5 --warmup 2 --cpu-frames 0 --no-side > $O/bench_b1.json 2> /dev/null ) ; echo "bench bf16 rc=$?" >> $O/log
( timeout 300 python bench.py --steps 5 --warmup 2 --cpu-frames 0 --no-side --dtype fp8w > $O/bench_b1_fp8w.json 2> /dev/null ) ; echo "bench fp8w rc=$?" >> $O/log
( timeout 300 python bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-side --dtype fp32 > $O/bench_b1_fp32.json 2> /dev/null ) ; echo "bench fp32 rc=$?" >> $O/log
cat $O/log; tail -2 $O/pytest_persist.log
python - <<'PY'
import json
for f in ('bench_b1','bench_b1_fp8w','bench_b1_fp32'):
    try:
        r=json.loads(open(f'gpurun_out/r5o/{f}.json').read().strip().split('\n')[-1])
        print(f, {k:r[k] for k in ('value','ms_per_step','phase_ms')}, r['roofline']['step_us'], r['roofline']['frac'], r['roofline'].get('traffic_per_step'), r['config'].get('persist'))
    except Exception as e: print(f, 'parse', e)
PY
