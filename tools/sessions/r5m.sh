#!/bin/bash
# Round 5, GPU session M: first-sweep waits for the fp32 and fp8-weight forms of the persistent launch.
O=gpurun_out/r5m; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python tools/fresh_box_probe.py --out $O/first > $O/first.out 2> $O/first.err ) ; echo "first-process probe rc=$?" >> $O/log
for DT in fp32 fp8w; do
( timeout 600 python tools/persist_probe.py --dtype $DT --out $O/probe_$DT --steps 400 --rounds 2 --skip-check --trace --variants \
   "pf=3" "pf=3,naps=0x436867" "pf=3,naps=0x214645" "pf=3,naps=0x327756" "pf=3,naps=0x325766" "pf=3,naps=0x325757" "pf=3,naps=0x335854" "pf=3,naps=0x325856" "pf=3,naps=0x425756" "pf=3,naps=0x324756" "pf=0" \
   > $O/probe_$DT.log 2>&1 ) ; echo "probe $DT rc=$?" >> $O/log
grep "\[time\]" $O/probe_$DT.log | tail -1 | python3 -c "
import sys,json
l=sys.stdin.read().split('[time] ')[1]
d=json.loads(l)
for k,v in sorted(d.items(), key=lambda kv: sum(kv[1])/len(kv[1])): print('$DT', k, v, round(sum(v)/len(v),2))
"
done
cat $O/log
