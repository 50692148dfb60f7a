#!/bin/bash
# Round 5, GPU session Q: the whole GPU suite + smoke() on the final tree (after the per-mode waits, the sanitizer target and the
# VLE_LIB override), first process = the instrumented probe as in every session.
O=gpurun_out/r5q; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python tools/fresh_box_probe.py --out $O/first > $O/first.out 2> $O/first.err ) ; echo "first-process probe rc=$?" >> $O/log
( timeout 900 python -m pytest tests -q -m gpu -x > $O/tests_all.log 2>&1 ) ; echo "pytest -m gpu rc=$?" >> $O/log
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 ) ; echo "smoke rc=$?" >> $O/log
cat $O/log; tail -3 $O/tests_all.log; tail -3 $O/smoke.log
