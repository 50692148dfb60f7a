# round 2, call B: re-run the new parity / bench tests, the in-kernel timeline + steps-per-graph sweep, host_prog A/B
cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_sizes_gpu.py tests/test_bench_gpu.py tests/test_engine_gpu.py -q -s -x > $D/tests_new.log 2>&1; echo "tests rc=$?"; grep -E "C2 full|forced batch|passed|failed|Error" $D/tests_new.log | tail -n 20
timeout 600 python tools/ktrace_step.py --out $D/ktrace_hostprog1 > $D/ktrace1.log 2>&1; echo "ktrace rc=$?"; tail -n 3 $D/ktrace1.log | cut -c1-1500
timeout 300 python bench.py --steps 5 --warmup 2 --cpu-frames 0 --no-c3 > $D/bench_hp1.log 2>&1; tail -n 1 $D/bench_hp1.log | cut -c1-400
timeout 300 python bench.py --steps 5 --warmup 2 --cpu-frames 0 --no-c3 --opt host_prog=0 > $D/bench_hp0.log 2>&1; tail -n 1 $D/bench_hp0.log | cut -c1-400
timeout 300 python bench.py --steps 5 --warmup 2 --cpu-frames 0 --no-c3 --opt steps_per_graph=32 > $D/bench_spg32.log 2>&1; tail -n 1 $D/bench_spg32.log | cut -c1-400
timeout 300 python bench.py --steps 3 --warmup 1 --cpu-frames 0 --batch 64 > $D/bench_b64.log 2>&1; tail -n 1 $D/bench_b64.log | cut -c1-500
