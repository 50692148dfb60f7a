cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $D/tests.log 2>&1; echo "tests rc=$?"; tail -n 3 $D/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $D/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $D/smoke.log
SECONDS=0; timeout 900 python bench.py > $D/bench_default.log 2>&1; echo "default bench rc=$? wall ${SECONDS}s"; tail -n 1 $D/bench_default.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['phase_ms'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_source'][:60])
for k in ('c3_batch64','fp32_exact','c5_share_fp8'):
    o=d.get(k,{}); print(k, o.get('value'), o.get('phase_ms'), (o.get('roofline_ar') or {}).get('frac'), (o.get('roofline_nar') or {}).get('frac'), o.get('error'))
"
