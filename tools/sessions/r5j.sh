#!/bin/bash
# Round 5, GPU session J: the persistent launch on fp8 weight rows (FP8W): tests, bench leg.
O=gpurun_out/r5j; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python tools/fresh_box_probe.py --out $O/first > $O/first.out 2> $O/first.err ) ; echo "first-process probe rc=$?" >> $O/log
( timeout 900 python -m pytest tests/test_persist_gpu.py -x -q -m gpu > $O/pytest_persist.log 2>&1 ) ; echo "pytest persist rc=$?" >> $O/log
( timeout 900 python -m pytest tests/test_fp8w_gpu.py tests/test_fp8_gpu.py tests/test_parity_sizes_gpu.py -x -q -m gpu > $O/pytest_fp8.log 2>&1 ) ; echo "pytest fp8 rc=$?" >> $O/log
( timeout 300 python bench.py --steps 5 --warmup 2 --cpu-frames 0 --no-side --dtype fp8w > $O/bench_b1_fp8w.json 2> $O/bench_b1_fp8w.err ) ; echo "bench fp8w rc=$?" >> $O/log
( timeout 300 python bench.py --steps 5 --warmup 2 --cpu-frames 0 --no-side --dtype fp8w --opt persist=0 > $O/bench_b1_fp8w_chain.json 2> /dev/null ) ; echo "bench fp8w chain rc=$?" >> $O/log
cat $O/log; tail -15 $O/pytest_persist.log; tail -3 $O/pytest_fp8.log
python - <<'PY'
import json
for f in ('bench_b1_fp8w','bench_b1_fp8w_chain'):
    try:
        r=json.loads(open(f'gpurun_out/r5j/{f}.json').read().strip().split('\n')[-1])
        print(f, {k:r[k] for k in ('value','ms_per_step','phase_ms')}, r['roofline']['step_us'], r['roofline']['frac'], r['config'].get('persist'))
    except Exception as e: print(f, 'parse', e)
PY
