#!/bin/bash
# Round 5, GPU session F: D2 as the default -- first-sweep waits re-timed (persist_naps), iterations per launch, timeline; bench line.
O=gpurun_out/r5f; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python tools/fresh_box_probe.py --out $O/first > $O/first.out 2> $O/first.err ) ; echo "first-process probe rc=$?" >> $O/log
# naps nibbles (low to high): attention output, x, x', hidden, q/k/v, partials; default 0x335854
( timeout 900 python tools/persist_probe.py --out $O/probe --steps 500 --rounds 2 --skip-check --variants \
   "pf=3" "pf=3,naps=0x335754" "pf=3,naps=0x335954" "pf=3,naps=0x335844" "pf=3,naps=0x335864" "pf=3,naps=0x335853" "pf=3,naps=0x335855" \
   "pf=3,naps=0x334854" "pf=3,naps=0x336854" "pf=3,naps=0x325854" "pf=3,naps=0x345854" "pf=3,naps=0x235854" "pf=3,naps=0x435854" "pf=3,naps=0x535854" \
   "pf=3,steps=64" "pf=3,steps=128" --trace "pf=3" > $O/probe.log 2>&1 ) ; echo "probe rc=$?" >> $O/log
( timeout 600 python bench.py --no-c3 --no-c5 --no-fp32 > $O/bench.json 2> $O/bench.err ) ; echo "bench rc=$?" >> $O/log
cat $O/log; grep "\[time\]" $O/probe.log | tail -1
python - <<'PY'
import json
try:
    r=json.loads(open('gpurun_out/r5f/bench.json').read().strip().split('\n')[-1])
    print({k:r[k] for k in ('value','ms_per_step','phase_ms')}, r['roofline']['step_us'], r['roofline']['frac'], r['config'].get('persist'))
except Exception as e: print('bench parse', e)
PY
