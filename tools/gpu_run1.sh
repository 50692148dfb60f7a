set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r1b
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/r1b/tests.log 2>&1; echo "tests rc=$?"
tail -n 15 gpurun_out/r1b/tests.log
timeout 300 python tools/ar_tune.py --steps 200 --rounds 3 > gpurun_out/r1b/tune.log 2>&1; echo "tune rc=$?"
tail -n 3 gpurun_out/r1b/tune.log
timeout 400 python bench.py --steps 5 --warmup 2 > gpurun_out/r1b/bench.log 2>&1; echo "bench rc=$?"
tail -n 2 gpurun_out/r1b/bench.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r1b/prof -o b1 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-frames 0 > $GRAFT_REPO_ROOT/gpurun_out/r1b/prof_run.log 2>&1); echo "prof rc=$?"
ls -R gpurun_out/r1b/prof | head -20
