cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
for v in 0 38000 30000 19000 15000; do
  timeout 300 python bench.py --batch 64 --steps 2 --warmup 1 --cpu-frames 0 --no-c3 --no-fp32 --opt attn_lds_pad=$v > $D/b64_$v.log 2>&1; tail -n 1 $D/b64_$v.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lds_pad=$v', d['value'], d['phase_ms'], d['roofline']['launch_us'])"
done
timeout 600 python tools/ktrace_step.py --out $D/ktrace_b64 --spg 8 --batch 64 --opt attn_lds_pad=38000 > $D/ktrace_b64.log 2>&1; head -6 $D/ktrace_b64_timeline.csv
