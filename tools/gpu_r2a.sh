# round 2, call A: the new parity tests + whole GPU suite, smoke, and the timestamped kernel trace of the graph-replayed AR step
cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_sizes_gpu.py tests/test_bench_gpu.py -q -s > $D/tests_new.log 2>&1; echo "new tests rc=$?"; tail -n 30 $D/tests_new.log
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_parity_sizes_gpu.py --deselect tests/test_bench_gpu.py > $D/tests_old.log 2>&1; echo "old tests rc=$?"; tail -n 8 $D/tests_old.log
python -c "import __graft_entry__ as g; g.smoke()" > $D/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 4 $D/smoke.log
(cd /tmp && rm -rf /tmp/tr1 && timeout 400 rocprofv3 --kernel-trace -d /tmp/tr1 -o t --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --cpu-frames 0 --no-c3 > $GRAFT_REPO_ROOT/$D/trace_bench.log 2>&1); echo "trace rc=$?"
F=$(find /tmp/tr1 -name "*kernel_trace.csv" | head -1); ls -la $F
python tools/trace_step.py $F $D/b1_graph --steps 8
tail -n 1 $D/trace_bench.log | cut -c1-600
