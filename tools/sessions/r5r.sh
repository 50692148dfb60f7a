#!/bin/bash
# Round 5, GPU session R: first process = the probe with every engine allocation and the caller's buffers in guarded mappings;
# then the default bench line of the final tree (all side legs).
O=gpurun_out/r5r; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python tools/fresh_box_probe.py --guard 1 --out $O/first > $O/first.out 2> $O/first.err ) ; echo "first-process probe (guard 1) rc=$?" >> $O/log
( timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) ; echo "bench default rc=$?" >> $O/log
cat $O/log; tail -c 3000 $O/bench_default.json
