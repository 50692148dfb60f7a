cd $GRAFT_REPO_ROOT
D=gpurun_out/r6l; mkdir -p $D
export TMPDIR=/tmp
timeout 300 python tools/fresh_box_probe.py --out $D/first > $D/probe.log 2>&1; echo "probe rc=$?"
timeout 900 python bench.py > $D/bench_default.log 2> $D/bench_default.err; echo "default bench rc=$?"; tail -n 1 $D/bench_default.log | cut -c1-260
timeout 600 python -m pytest tests/test_bench_gpu.py -x -q > $D/tests.log 2>&1; echo "tests rc=$?"; tail -n 2 $D/tests.log
for b in 3 8; do timeout 300 python bench.py --batch $b --no-side --cpu-frames 0 --steps 3 --warmup 1 > $D/bench_b$b.log 2>&1; tail -n 1 $D/bench_b$b.log | cut -c1-200; done
