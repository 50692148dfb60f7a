# One gpurun call that reproduces what profiles/ holds for a session:   gpurun -- 'bash tools/gpu_ci.sh <tag>'
# GPU tests, the default bench line (B = 1 + cpu_baseline + c3_batch64), rocprofv3 kernel stats at B = 1 and B = 64.
cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > $D/tests.log 2>&1; echo "tests rc=$?"; tail -n 4 $D/tests.log
timeout 600 python bench.py > $D/bench_default.log 2>&1; echo "default bench rc=$?"; tail -n 1 $D/bench_default.log
(cd /tmp && rm -rf /tmp/prof1 && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof1 -o b1 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-side > $GRAFT_REPO_ROOT/$D/prof1.log 2>&1); echo "prof1 rc=$?"
cp /tmp/prof1/b1_kernel_stats.csv $D/ 2>/dev/null
(cd /tmp && rm -rf /tmp/prof64 && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof64 -o b64 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --batch 64 --steps 1 --warmup 0 --cpu-frames 0 > $GRAFT_REPO_ROOT/$D/prof64.log 2>&1); echo "prof64 rc=$?"
cp /tmp/prof64/b64_kernel_stats.csv $D/ 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" > $D/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $D/smoke.log
