cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["phase_ms"], d["roofline"]["launch_us"], d["roofline"]["frac"])'
timeout 300 python tools/gemm_bench.py --quick > $D/gemm_bench.log 2>&1; cat $D/gemm_bench.log
for v in 0 1 3; do
timeout 300 python bench.py --batch 64 --steps 2 --warmup 1 --cpu-frames 0 --no-side --opt g8_nt=$v > $D/b64_$v.log 2>&1; tail -n 1 $D/b64_$v.log | python -c "$P" b64_g8nt=$v
done
