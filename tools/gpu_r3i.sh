cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["phase_ms"], d["roofline"]["launch_us"], d["roofline"]["frac"])'
timeout 300 python bench.py --batch 64 --steps 2 --warmup 1 --cpu-frames 0 --no-c3 --no-fp32 --opt gs_formal=1 > $D/b64_formal.log 2>&1; tail -n 1 $D/b64_formal.log | python -c "$P" b64_formal
timeout 300 python bench.py --batch 64 --steps 2 --warmup 1 --cpu-frames 0 --no-c3 --no-fp32 > $D/b64.log 2>&1; tail -n 1 $D/b64.log | python -c "$P" b64
timeout 300 python bench.py --steps 5 --warmup 2 --cpu-frames 0 --no-c3 --no-fp32 --dtype fp8w > $D/b1_fp8w.log 2>&1; tail -n 1 $D/b1_fp8w.log | python -c "$P" b1_fp8w
timeout 300 python bench.py --batch 8 --steps 3 --warmup 1 --cpu-frames 0 --no-c3 --no-fp32 > $D/b8.log 2>&1; tail -n 1 $D/b8.log | python -c "$P" b8
/usr/bin/time -v timeout 900 python bench.py --steps 2 --warmup 1 --cpu-frames 0 --d-model 1536 --layers 24 --nhead 16 --dtype fp8 --batch 32 --no-c3 --no-fp32 > $D/c5_fp8.log 2> $D/c5_time.log; tail -n 1 $D/c5_fp8.log | python -c "$P" c5_fp8; grep -E "Elapsed|Maximum resident" $D/c5_time.log
