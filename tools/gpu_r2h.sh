# round 2, call H: A/B of the gemm_skinny X-rotation knob (prefetched epilogue operands kept, G = 4)
cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
for R in 0 1; do
timeout 300 python bench.py --steps 2 --warmup 1 --cpu-frames 0 --batch 64 --opt gs_rot=$R > $D/bench_b64_rot$R.log 2>&1; echo "b64 rot=$R $(tail -n 1 $D/bench_b64_rot$R.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["phase_ms"], d["roofline"]["launch_us"], d["roofline"]["frac"])')"
timeout 300 python bench.py --steps 2 --warmup 1 --cpu-frames 0 --batch 8 --opt gs_rot=$R > $D/bench_b8_rot$R.log 2>&1; echo "b8 rot=$R $(tail -n 1 $D/bench_b8_rot$R.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["phase_ms"], d["roofline"]["launch_us"])')"
timeout 600 python bench.py --steps 1 --warmup 1 --cpu-frames 0 --d-model 1536 --layers 24 --nhead 16 --dtype fp8 --batch 32 --opt gs_rot=$R > $D/bench_c5_rot$R.log 2>&1; echo "c5 fp8 rot=$R $(tail -n 1 $D/bench_c5_rot$R.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["phase_ms"], d["roofline"]["launch_us"], d["roofline"]["frac"])')"
done
