# round 6, session f: two utterances decoded one after the other on the persistent launch
cd $GRAFT_REPO_ROOT
D=gpurun_out/r6f; mkdir -p $D
export TMPDIR=/tmp
timeout 300 python tools/fresh_box_probe.py --out $D/first > $D/probe.log 2>&1; echo "probe rc=$?"
timeout 900 python -m pytest tests/test_persist_gpu.py tests/test_bench_gpu.py -x -q > $D/tests.log 2>&1; echo "tests rc=$?"; tail -n 5 $D/tests.log
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -k "ragged or batch" > $D/tests2.log 2>&1; echo "tests2 rc=$?"; tail -n 3 $D/tests2.log
timeout 300 python bench.py --batch 2 --no-side --cpu-frames 0 --steps 4 --warmup 1 > $D/bench_b2.log 2>&1; tail -n 1 $D/bench_b2.log | cut -c1-900
