# round 2, call D: fused LayerNorm in the batched AR step -- parity + A/B timing
cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_fp8w_gpu.py tests/test_serving_gpu.py tests/test_parity_sizes_gpu.py -q -x -k "fused or batch or fp8w or serving or c3 or c5 or slot" > $D/tests.log 2>&1; echo "tests rc=$?"; tail -n 12 $D/tests.log
for F in 1 0; do timeout 300 python bench.py --steps 2 --warmup 1 --cpu-frames 0 --batch 64 --opt gs_fuse_ln=$F > $D/bench_b64_fuse$F.log 2>&1; echo "b64 fuse=$F $(tail -n 1 $D/bench_b64_fuse$F.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["phase_ms"], d["roofline"]["launch_us"], d["roofline"]["frac"])')"; done
for F in 1 0; do timeout 300 python bench.py --steps 2 --warmup 1 --cpu-frames 0 --batch 8 --opt gs_fuse_ln=$F > $D/bench_b8_fuse$F.log 2>&1; echo "b8 fuse=$F $(tail -n 1 $D/bench_b8_fuse$F.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["phase_ms"], d["roofline"]["launch_us"])')"; done
timeout 300 python bench.py --steps 2 --warmup 1 --cpu-frames 0 --batch 64 --profile-kernels 64 > $D/bench_b64_prof.log 2>&1; tail -n 1 $D/bench_b64_prof.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["roofline"].get("kernel_us"))'
