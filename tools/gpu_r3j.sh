cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["phase_ms"], d["roofline"]["launch_us"], d["roofline"]["frac"])'
nproc
python - <<'PY' > $D/c5_load.log 2>&1
import time, torch, valle_amd, bench
t0=time.time(); torch.manual_seed(0)
m=valle_amd.VALLE(1536,16,24,prefix_mode=1,engine_dtype="fp8",max_batch=32); t1=time.time()
m=m.to("cuda").eval(); t2=time.time()
eng=m.engine_for(32, bench.S_TEXT, bench.P_PROMPT); t3=time.time()
print("init %.1f s, to(cuda) %.1f s, engine load (state_dict -> quantise -> upload -> pack) %.1f s"%(t1-t0,t2-t1,t3-t2))
PY
cat $D/c5_load.log
SECONDS=0
timeout 900 python bench.py --steps 2 --warmup 1 --cpu-frames 0 --d-model 1536 --layers 24 --nhead 16 --dtype fp8 --batch 32 --no-c3 --no-fp32 > $D/c5_fp8.log 2>&1; tail -n 1 $D/c5_fp8.log | python -c "$P" c5_fp8; echo "c5 bench wall ${SECONDS}s"
