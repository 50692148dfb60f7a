#!/bin/bash
# Round 5, GPU session K: the persistent launch in fp32 (token-exact mode) -- does it load and run, tests, the fp32 bench leg.
O=gpurun_out/r5k; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python tools/fresh_box_probe.py --out $O/first > $O/first.out 2> $O/first.err ) ; echo "first-process probe rc=$?" >> $O/log
( timeout 600 python -m pytest tests/test_persist_gpu.py -x -q -m gpu -k "fp32 or default_where" > $O/pytest_fp32.log 2>&1 ) ; echo "pytest fp32 persist rc=$?" >> $O/log
( timeout 900 python -m pytest tests/test_parity_sizes_gpu.py tests/test_engine_gpu.py -x -q -m gpu -k "fp32 or golden or c2_arch" > $O/pytest_exact.log 2>&1 ) ; echo "pytest exact rc=$?" >> $O/log
( timeout 300 python bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-side --dtype fp32 > $O/bench_b1_fp32.json 2> $O/bench_b1_fp32.err ) ; echo "bench fp32 rc=$?" >> $O/log
( timeout 300 python bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-side --dtype fp32 --opt persist=0 > $O/bench_b1_fp32_chain.json 2> /dev/null ) ; echo "bench fp32 chain rc=$?" >> $O/log
cat $O/log; tail -12 $O/pytest_fp32.log | cut -c1-200; tail -3 $O/pytest_exact.log; tail -2 $O/bench_b1_fp32.err | cut -c1-300
python - <<'PY'
import json
for f in ('bench_b1_fp32','bench_b1_fp32_chain'):
    try:
        r=json.loads(open(f'gpurun_out/r5k/{f}.json').read().strip().split('\n')[-1])
        print(f, {k:r[k] for k in ('value','ms_per_step','phase_ms')}, r['roofline']['step_us'], r['roofline']['frac'], r['config'].get('persist'))
    except Exception as e: print(f, 'parse', e)
PY
