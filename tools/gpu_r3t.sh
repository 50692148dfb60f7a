cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["phase_ms"], d["roofline"]["launch_us"], d["roofline"]["frac"])'
timeout 300 python bench.py --batch 64 --steps 2 --warmup 1 --cpu-frames 0 --no-side > $D/a.log 2>&1; tail -n 1 $D/a.log | python -c "$P" default
timeout 300 python bench.py --batch 64 --steps 2 --warmup 1 --cpu-frames 0 --no-side --opt gs_msplit=2 > $D/b.log 2>&1; tail -n 1 $D/b.log | python -c "$P" msplit2
timeout 300 python bench.py --batch 64 --steps 2 --warmup 1 --cpu-frames 0 --no-side --opt gs_msplit=2 --opt gs_ms_pad=98304 > $D/c.log 2>&1; tail -n 1 $D/c.log | python -c "$P" msplit2_pad96k
