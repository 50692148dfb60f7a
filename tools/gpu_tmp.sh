cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_ops_gpu.py tests/test_modules_gpu.py -m gpu -x -q -k "attention or multihead or encoder" > $D/tests_attn.log 2>&1; echo "attn tests rc=$?"; tail -n 3 $D/tests_attn.log | cut -c1-300
for o in "attn_qw=1" "attn_qw=2" "attn_qw=1" "attn_qw=2"; do
  timeout 300 python bench.py --batch 64 --steps 1 --warmup 1 --cpu-frames 0 --opt $o > $D/bench_b64_$o.log 2>&1; echo "b64 $o rc=$?"; tail -n 1 $D/bench_b64_$o.log | grep -o '"value": [0-9.]*\|"phase_ms[^}]*}'
done
for o in "nsplit=8" "steps_per_graph=16" "nsplit=4"; do
  timeout 300 python bench.py --steps 3 --warmup 1 --cpu-frames 0 --opt $o > $D/bench_b1_$o.log 2>&1; echo "b1 $o rc=$?"; tail -n 1 $D/bench_b1_$o.log | grep -o '"value": [0-9.]*\|"phase_ms[^}]*}'
done
