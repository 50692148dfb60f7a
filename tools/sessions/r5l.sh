#!/bin/bash
# Round 5, GPU session L: the whole GPU suite with every engine allocation poisoned (0xff), and the engine / persist / parity tests with
# every engine allocation at the edge of its own mapping (VLE_GUARD_ALLOC=1).
O=gpurun_out/r5l; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python tools/fresh_box_probe.py --out $O/first --poison 0x7f --guard 1 > $O/first.out 2> $O/first.err ) ; echo "first-process probe (poison 7f + guard 1) rc=$?" >> $O/log
( VLE_POISON_ALLOC=0xff timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_poison_all.log 2>&1 ) ; echo "pytest all under poison rc=$?" >> $O/log
( VLE_GUARD_ALLOC=1 timeout 900 python -m pytest tests/test_persist_gpu.py tests/test_engine_gpu.py -q -m gpu -x > $O/pytest_guard.log 2> $O/pytest_guard.err ) ; echo "pytest persist+engine under guard rc=$?" >> $O/log
cat $O/log; tail -4 $O/pytest_poison_all.log; tail -4 $O/pytest_guard.log
