#!/bin/bash
# Round 5, GPU session E: reworked LayerNorm fold (coalesced statistics, sg / tb through LDS) A/B + profile; D2 default; bench line; full suite.
O=gpurun_out/r5e; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python tools/fresh_box_probe.py --out $O/first > $O/first.out 2> $O/first.err ) ; echo "first-process probe rc=$?" >> $O/log
( timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py tests/test_persist_gpu.py -x -q -m gpu -k "layernorm or persist or caller or dot2" > $O/pytest_new.log 2>&1 ) ; echo "pytest new rc=$?" >> $O/log
( timeout 300 python tools/nar_ab.py --batch 1 --reps 4 --opt ln_fold=0 --opt ln_fold=1 > $O/ab_b1.json 2> $O/ab_b1.err ) ; echo "ab b1 rc=$?" >> $O/log
( timeout 400 python tools/nar_ab.py --batch 64 --reps 2 --steps 8 --opt ln_fold=0 --opt ln_fold=1 > $O/ab_b64.json 2> $O/ab_b64.err ) ; echo "ab b64 rc=$?" >> $O/log
( cd /tmp && rm -rf /tmp/pp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pp -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/nar_ab.py --batch 1 --reps 3 --opt ln_fold=1 > $GRAFT_REPO_ROOT/$O/prof_b1_fold1.out 2>&1 ; cp $(find /tmp/pp -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/b1_fold1_kernel_stats.csv ) ; echo "prof b1 fold1 rc=$?" >> $O/log
( cd /tmp && rm -rf /tmp/pp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pp -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/nar_ab.py --batch 64 --reps 1 --steps 8 --opt ln_fold=1 > $GRAFT_REPO_ROOT/$O/prof_b64_fold1.out 2>&1 ; cp $(find /tmp/pp -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/b64_fold1_kernel_stats.csv ) ; echo "prof b64 fold1 rc=$?" >> $O/log
( timeout 600 python bench.py > $O/bench.json 2> $O/bench.err ) ; echo "bench rc=$?" >> $O/log
( timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1 ) ; echo "pytest all rc=$?" >> $O/log
cat $O/log; tail -12 $O/pytest_new.log; cat $O/ab_b1.json $O/ab_b64.json; tail -5 $O/pytest_all.log
python - <<'PY'
import json
try:
    r=json.loads(open('gpurun_out/r5e/bench.json').read().strip().split('\n')[-1])
    print({k:r[k] for k in ('value','ms_per_step','phase_ms')}, r['roofline']['step_us'], r['roofline']['frac'], r['config'].get('persist'))
    for k in ('c3_batch64','sampled','s200','fp32_exact','c5_share_fp8w'):
        if k in r: print(k, r[k].get('value'), r[k].get('phase_ms'), r[k].get('persist'), r[k].get('roofline_ar',{}).get('frac'), r[k].get('roofline_nar',{}).get('frac'))
except Exception as e: print('bench parse', e)
PY
