#!/bin/bash
# Round 5, GPU session I: batched-step LN consumer with v_rcp / v_rsq (tests), stress of the persistent forms, the C2 full-size parity numbers, attention block width at one utterance.
O=gpurun_out/r5i; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python tools/fresh_box_probe.py --out $O/first > $O/first.out 2> $O/first.err ) ; echo "first-process probe rc=$?" >> $O/log
( timeout 900 python -m pytest tests/test_parity_sizes_gpu.py -x -q -m gpu -s -k "c2_full_size_bf16 or c3" > $O/pytest_sizes.log 2>&1 ) ; echo "pytest sizes rc=$?" >> $O/log
( timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_fp8w_gpu.py tests/test_serving_gpu.py -x -q -m gpu > $O/pytest_engine.log 2>&1 ) ; echo "pytest engine rc=$?" >> $O/log
( timeout 200 python tools/persist_stress.py 40 > $O/persist_stress.log 2>&1 ) ; echo "stress rc=$?" >> $O/log
( timeout 300 python tools/nar_ab.py --batch 1 --reps 3 --opt attn_q128=0 --opt attn_q128=1 > $O/ab_q128.json 2> $O/ab_q128.err ) ; echo "ab q128 rc=$?" >> $O/log
( timeout 400 python bench.py --batch 64 --steps 2 --warmup 1 --cpu-frames 0 --no-side > $O/bench_b64.json 2> $O/bench_b64.err ) ; echo "bench b64 rc=$?" >> $O/log
cat $O/log; grep -h "C2 full\|passed\|failed" $O/pytest_sizes.log | tail -5; tail -2 $O/pytest_engine.log; tail -1 $O/persist_stress.log; cat $O/ab_q128.json; tail -1 $O/bench_b64.json | cut -c1-400
