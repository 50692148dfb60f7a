#!/bin/bash
# Round 5, GPU session H: LayerNorm fold with the cheap epilogue (v_rsq, hoisted LDS reads) A/B; persist default with the re-timed waits.
O=gpurun_out/r5h; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python tools/fresh_box_probe.py --out $O/first > $O/first.out 2> $O/first.err ) ; echo "first-process probe rc=$?" >> $O/log
( timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py -x -q -m gpu -k "layernorm" > $O/pytest_ln.log 2>&1 ) ; echo "pytest ln rc=$?" >> $O/log
( timeout 300 python tools/nar_ab.py --batch 1 --reps 4 --opt ln_fold=0 --opt ln_fold=1 > $O/ab_b1.json 2> $O/ab_b1.err ) ; echo "ab b1 rc=$?" >> $O/log
( timeout 400 python tools/nar_ab.py --batch 64 --reps 2 --steps 8 --opt ln_fold=0 --opt ln_fold=1 > $O/ab_b64.json 2> $O/ab_b64.err ) ; echo "ab b64 rc=$?" >> $O/log
( cd /tmp && rm -rf /tmp/pp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pp -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/nar_ab.py --batch 1 --reps 3 --opt ln_fold=1 > $GRAFT_REPO_ROOT/$O/prof_b1_fold1.out 2>&1 ; cp $(find /tmp/pp -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/b1_fold1_kernel_stats.csv ) ; echo "prof b1 fold1 rc=$?" >> $O/log
( cd /tmp && rm -rf /tmp/pp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pp -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/nar_ab.py --batch 64 --reps 1 --steps 8 --opt ln_fold=1 > $GRAFT_REPO_ROOT/$O/prof_b64_fold1.out 2>&1 ; cp $(find /tmp/pp -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/b64_fold1_kernel_stats.csv ) ; echo "prof b64 fold1 rc=$?" >> $O/log
( timeout 600 python bench.py --no-c3 --no-c5 --no-fp32 > $O/bench.json 2> $O/bench.err ) ; echo "bench rc=$?" >> $O/log
cat $O/log; tail -3 $O/pytest_ln.log; cat $O/ab_b1.json $O/ab_b64.json
python - <<'PY'
import json
try:
    r=json.loads(open('gpurun_out/r5h/bench.json').read().strip().split('\n')[-1])
    print({k:r[k] for k in ('value','ms_per_step','phase_ms')}, r['roofline']['step_us'], r['roofline']['frac'], r['config'].get('persist'))
except Exception as e: print('bench parse', e)
PY
