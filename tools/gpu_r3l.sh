cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["phase_ms"], d["roofline"]["launch_us"], d["roofline"]["frac"])'
for v in 0 1 -1; do
timeout 300 python bench.py --batch 64 --steps 2 --warmup 1 --cpu-frames 0 --no-side --opt attn_nt=$v > $D/b64_$v.log 2>&1; tail -n 1 $D/b64_$v.log | python -c "$P" b64_nt=$v
done
timeout 300 python bench.py --batch 8 --steps 3 --warmup 1 --cpu-frames 0 --no-side --opt attn_nt=1 > $D/b8.log 2>&1; tail -n 1 $D/b8.log | python -c "$P" b8_nt1
timeout 300 python bench.py --batch 8 --steps 3 --warmup 1 --cpu-frames 0 --no-side --opt attn_nt=0 > $D/b8.log 2>&1; tail -n 1 $D/b8.log | python -c "$P" b8_nt0
