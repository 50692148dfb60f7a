cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "decode or skinny" > $D/tests.log 2>&1; echo "tests rc=$?"; tail -n 3 $D/tests.log
timeout 300 python tools/ar_tune.py --steps 300 --rounds 2 2>&1 | tail -n 1 | tee $D/tune.log
timeout 300 python tools/op_chain_bench.py > $D/opchain.log 2>&1; tail -n 20 $D/opchain.log
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $GRAFT_REPO_ROOT/$D/pmc_fetch -o f --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --cpu-frames 0 --no-graph > $GRAFT_REPO_ROOT/$D/pmc_fetch.log 2>&1); echo "pmc rc=$?"
(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $GRAFT_REPO_ROOT/$D/pmc_write -o w --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --cpu-frames 0 --no-graph > $GRAFT_REPO_ROOT/$D/pmc_write.log 2>&1); echo "pmc rc=$?"
ls -la $D/pmc_fetch $D/pmc_write | head -20
