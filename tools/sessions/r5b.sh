#!/bin/bash
# Round 5, GPU session B: LayerNorm fold + untraced persistent step: tests, poison re-run, bench line.
O=gpurun_out/r5b; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python tools/fresh_box_probe.py --out $O/first > $O/first.out 2> $O/first.err ) ; echo "first-process probe rc=$?" >> $O/log
( VLE_POISON_ALLOC=0xff timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "fused_layernorm or layernorm_folded" > $O/pytest_poison.log 2>&1 ) ; echo "pytest poison rc=$?" >> $O/log
( timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py tests/test_persist_gpu.py -x -q -m gpu -k "layernorm or persist or caller" > $O/pytest_new.log 2>&1 ) ; echo "pytest new rc=$?" >> $O/log
( timeout 600 python bench.py > $O/bench.json 2> $O/bench.err ) ; echo "bench rc=$?" >> $O/log
( timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1 ) ; echo "pytest all rc=$?" >> $O/log
cat $O/log
tail -5 $O/pytest_poison.log; tail -15 $O/pytest_new.log; tail -5 $O/pytest_all.log
cat $O/first.out
python - <<'PY'
import json
try:
    r=json.loads(open('gpurun_out/r5b/bench.json').read().strip().split('\n')[-1])
    print({k:r[k] for k in ('value','ms_per_step','phase_ms')}, r['roofline']['step_us'], r['roofline']['frac'], r['config'].get('persist'))
    for k in ('c3_batch64','sampled','s200','fp32_exact'):
        if k in r: print(k, r[k].get('value'), r[k].get('phase_ms'), r[k].get('persist'))
except Exception as e: print('bench parse', e)
PY
