#!/bin/bash
# Round 5, GPU session N: finer first-sweep waits per engine mode.
O=gpurun_out/r5n; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python tools/fresh_box_probe.py --out $O/first > $O/first.out 2> $O/first.err ) ; echo "first-process probe rc=$?" >> $O/log
run() { DT=$1; shift
( timeout 600 python tools/persist_probe.py --dtype $DT --out $O/probe_$DT --steps 400 --rounds 2 --skip-check --trace --variants "$@" > $O/probe_$DT.log 2>&1 ) ; echo "probe $DT rc=$?" >> $O/log
grep "\[time\]" $O/probe_$DT.log | tail -1 | python3 -c "
import sys,json
l=sys.stdin.read().split('[time] ')[1]
d=json.loads(l)
for k,v in sorted(d.items(), key=lambda kv: sum(kv[1])/len(kv[1])): print('$DT', k, v, round(sum(v)/len(v),2))
"; }
run bf16 "pf=3" "pf=3,naps=0x214645" "pf=3,naps=0x214646" "pf=3,naps=0x224646" "pf=3,naps=0x214656" "pf=3,naps=0x215646" "pf=3,naps=0x314646" "pf=3,naps=0x213646" "pf=3,naps=0x214636"
run fp8w "pf=3,naps=0x214645" "pf=3,naps=0x103534" "pf=3,naps=0x214535" "pf=3,naps=0x204645" "pf=3,naps=0x213645" "pf=3,naps=0x214644" "pf=3,naps=0x114645" "pf=3,naps=0x214635"
run fp32 "pf=3,naps=0x327756" "pf=3,naps=0x328756" "pf=3,naps=0x329756" "pf=3,naps=0x327746" "pf=3,naps=0x216645" "pf=3,naps=0x217645" "pf=3,naps=0x327656" "pf=3,naps=0x227756"
cat $O/log
