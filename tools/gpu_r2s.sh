cd $GRAFT_REPO_ROOT
D=gpurun_out/r2s; mkdir -p $D
timeout 600 python -m pytest tests/test_ops_gpu.py -q -k attention > $D/tests_attn.log 2>&1; echo "attn tests rc=$?"; tail -n 3 $D/tests_attn.log
timeout 400 python tools/attn_bench.py > $D/attn_bench5.log 2>&1; grep -E "^C|default|dma" $D/attn_bench5.log
timeout 300 python bench.py --steps 5 --warmup 2 --cpu-frames 0 --no-c3 > $D/bench_b1.log 2>&1; tail -n 1 $D/bench_b1.log | cut -c1-200; grep -o '"phase_ms": {[^}]*}' $D/bench_b1.log | tail -1
