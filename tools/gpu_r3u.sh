cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["phase_ms"], d["roofline"]["launch_us"], d["roofline"]["frac"])'
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "attention_mfma_long or attention_rows" > $D/t.log 2>&1; echo "test rc=$?"; tail -n 2 $D/t.log
timeout 300 python tools/attn_bench.py --lsum > $D/attn_lsum.log 2>&1; cat $D/attn_lsum.log
for v in 1 0; do
timeout 300 python bench.py --batch 64 --steps 2 --warmup 1 --cpu-frames 0 --no-side --opt attn_lsum=$v > $D/b64_$v.log 2>&1; tail -n 1 $D/b64_$v.log | python -c "$P" b64_lsum=$v
done
