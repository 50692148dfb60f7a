cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "linear_gemm or split_k" > $D/tests_new.log 2>&1; echo "gemm tests rc=$?"; tail -n 3 $D/tests_new.log
for o in "glds_big=0" "glds_big=-1" "glds_big=256"; do
  timeout 300 python bench.py --batch 64 --steps 1 --warmup 1 --cpu-frames 0 --opt $o > $D/bench_b64_$o.log 2>&1; echo "b64 $o rc=$?"; tail -n 1 $D/bench_b64_$o.log | cut -c1-120; tail -n 1 $D/bench_b64_$o.log | grep -o '"phase_ms[^}]*}'
done
timeout 300 python bench.py --batch 8 --steps 1 --warmup 1 --cpu-frames 0 > $D/bench_b8.log 2>&1; echo "b8 rc=$?"; tail -n 1 $D/bench_b8.log | cut -c1-120
# profiles: kernel-trace stats of the bench command (B = 1 and B = 64)
(cd /tmp && rm -rf /tmp/prof1 && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof1 -o b1 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-frames 0 > $GRAFT_REPO_ROOT/$D/prof1.log 2>&1); echo "prof1 rc=$?"
cp /tmp/prof1/b1_kernel_stats.csv $D/ 2>/dev/null
(cd /tmp && rm -rf /tmp/prof64 && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof64 -o b64 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --batch 64 --steps 1 --warmup 0 --cpu-frames 0 > $GRAFT_REPO_ROOT/$D/prof64.log 2>&1); echo "prof64 rc=$?"
cp /tmp/prof64/b64_kernel_stats.csv $D/ 2>/dev/null
head -n 8 $D/b1_kernel_stats.csv | cut -c1-100,150-260
timeout 600 python bench.py > $D/bench_default.log 2>&1; echo "default bench rc=$?"; tail -n 1 $D/bench_default.log
