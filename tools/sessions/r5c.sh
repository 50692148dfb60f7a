#!/bin/bash
# Round 5, GPU session C: where did the LayerNorm fold's time go (A/B + kernel profile), poison bisect, fixed persist test.
O=gpurun_out/r5c; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python tools/fresh_box_probe.py --out $O/first > $O/first.out 2> $O/first.err ) ; echo "first-process probe rc=$?" >> $O/log
( timeout 300 python tools/nar_ab.py --batch 1 --reps 4 --opt ln_fold=0 --opt ln_fold=1 > $O/ab_b1.json 2> $O/ab_b1.err ) ; echo "ab b1 rc=$?" >> $O/log
( timeout 400 python tools/nar_ab.py --batch 64 --reps 2 --steps 8 --opt ln_fold=0 --opt ln_fold=1 > $O/ab_b64.json 2> $O/ab_b64.err ) ; echo "ab b64 rc=$?" >> $O/log
for f in 0 1; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_b1_fold$f -- python $GRAFT_REPO_ROOT/tools/nar_ab.py --batch 1 --reps 2 --opt ln_fold=$f > $GRAFT_REPO_ROOT/$O/prof_b1_fold$f.out 2>&1 ) ; echo "prof b1 fold$f rc=$?" >> $O/log
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_b64_fold$f -- python $GRAFT_REPO_ROOT/tools/nar_ab.py --batch 64 --reps 1 --steps 8 --opt ln_fold=$f > $GRAFT_REPO_ROOT/$O/prof_b64_fold$f.out 2>&1 ) ; echo "prof b64 fold$f rc=$?" >> $O/log
done
for d in $O/prof_*; do [ -d $d ] && find $d -name "*kernel_stats.csv" -exec cp {} $d.kernel_stats.csv \; && rm -rf $d; done
( timeout 600 python tools/poison_bisect.py "tests/test_engine_gpu.py::test_fused_layernorm_batch_step_matches_layernorm_kernels[33-1536-16-2-fp8w]" > $O/bisect.log 2>&1 ) ; echo "bisect rc=$?" >> $O/log
( timeout 600 python -m pytest tests/test_persist_gpu.py -x -q -m gpu > $O/pytest_persist.log 2>&1 ) ; echo "pytest persist rc=$?" >> $O/log
cat $O/log; cat $O/ab_b1.json $O/ab_b64.json; tail -12 $O/bisect.log; tail -4 $O/pytest_persist.log
