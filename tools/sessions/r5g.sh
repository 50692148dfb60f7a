#!/bin/bash
# Round 5, GPU session G: D2 with the lane-parallel merge and the fast wave totals -- tests, first-sweep waits re-timed, bench line.
O=gpurun_out/r5g; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python tools/fresh_box_probe.py --out $O/first > $O/first.out 2> $O/first.err ) ; echo "first-process probe rc=$?" >> $O/log
( timeout 600 python -m pytest tests/test_persist_gpu.py -x -q -m gpu -s > $O/pytest_persist.log 2>&1 ) ; echo "pytest persist rc=$?" >> $O/log
# naps nibbles (low to high): attention output, x, x', hidden, q/k/v, partials; default 0x335854
( timeout 900 python tools/persist_probe.py --out $O/probe --steps 500 --rounds 2 --skip-check --variants \
   "pf=3" "pf=3,naps=0x335855" "pf=3,naps=0x335856" "pf=3,naps=0x335857" "pf=3,naps=0x325755" "pf=3,naps=0x325756" "pf=3,naps=0x335755" "pf=3,naps=0x325855" \
   "pf=3,naps=0x335845" "pf=3,naps=0x335865" "pf=3,naps=0x334855" "pf=3,naps=0x435855" "pf=3,naps=0x425755" "pf=3,naps=0x325745" "pf=3,naps=0x315755" \
   --trace "pf=3" > $O/probe.log 2>&1 ) ; echo "probe rc=$?" >> $O/log
( timeout 600 python bench.py --no-c3 --no-c5 --no-fp32 > $O/bench.json 2> $O/bench.err ) ; echo "bench rc=$?" >> $O/log
cat $O/log; tail -3 $O/pytest_persist.log; grep "D2 mode" $O/pytest_persist.log
grep "\[time\]" $O/probe.log | tail -1 | python3 -c "
import sys,json
l=sys.stdin.read().split('[time] ')[1]
d=json.loads(l)
for k,v in sorted(d.items(), key=lambda kv: sum(kv[1])/len(kv[1])): print(k, v, round(sum(v)/len(v),2))
"
python - <<'PY'
import json
try:
    r=json.loads(open('gpurun_out/r5g/bench.json').read().strip().split('\n')[-1])
    print({k:r[k] for k in ('value','ms_per_step','phase_ms')}, r['roofline']['step_us'], r['roofline']['frac'], r['config'].get('persist'))
except Exception as e: print('bench parse', e)
PY
