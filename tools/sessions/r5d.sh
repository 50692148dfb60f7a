#!/bin/bash
# Round 5, GPU session D: kernel profiles of the NAR passes with / without the LayerNorm fold; D2 persistent step (tests + timing).
O=gpurun_out/r5d; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python tools/fresh_box_probe.py --out $O/first > $O/first.out 2> $O/first.err ) ; echo "first-process probe rc=$?" >> $O/log
for f in 0 1; do
  ( cd /tmp && rm -rf /tmp/pp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pp -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/nar_ab.py --batch 1 --reps 3 --opt ln_fold=$f > $GRAFT_REPO_ROOT/$O/prof_b1_fold$f.out 2>&1 ; cp $(find /tmp/pp -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/b1_fold${f}_kernel_stats.csv ) ; echo "prof b1 fold$f rc=$?" >> $O/log
  ( cd /tmp && rm -rf /tmp/pp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pp -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/nar_ab.py --batch 64 --reps 1 --steps 8 --opt ln_fold=$f > $GRAFT_REPO_ROOT/$O/prof_b64_fold$f.out 2>&1 ; cp $(find /tmp/pp -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/b64_fold${f}_kernel_stats.csv ) ; echo "prof b64 fold$f rc=$?" >> $O/log
done
( timeout 600 python -m pytest tests/test_persist_gpu.py -x -q -m gpu > $O/pytest_persist.log 2>&1 ) ; echo "pytest persist rc=$?" >> $O/log
( timeout 600 python tools/persist_probe.py --out $O/probe --steps 400 --rounds 3 --skip-check --variants "pf=3" "pf=3,mode=0x174" "pf=3,mode=0x17c" "pf=3,mode=0x13c" --trace "pf=3" "pf=3,mode=0x174" > $O/probe.log 2>&1 ) ; echo "probe rc=$?" >> $O/log
cat $O/log; tail -4 $O/pytest_persist.log; grep "\[time\]" $O/probe.log | tail -1
