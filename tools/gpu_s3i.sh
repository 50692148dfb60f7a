cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_serving_gpu.py -m gpu -x -q > $D/tests_serving.log 2>&1; echo "serving tests rc=$?"; tail -n 3 $D/tests_serving.log | cut -c1-300
timeout 900 python tools/serve_bench.py --n 192 --max-batch 64 > $D/serve_bench.log 2>&1; echo "serve bench rc=$?"; grep -v amdgpu $D/serve_bench.log | tail -8
for o in "glds_prio=0" "glds_prio=1"; do
  timeout 300 python bench.py --batch 64 --steps 1 --warmup 1 --cpu-frames 0 --opt $o > $D/bench_b64_$o.log 2>&1; echo "b64 $o rc=$?"; tail -n 1 $D/bench_b64_$o.log | grep -o '"value": [0-9.]*\|"phase_ms[^}]*}'
done
timeout 300 python bench.py --steps 3 --warmup 1 --cpu-frames 0 --opt glds_prio=1 > $D/bench_b1_prio.log 2>&1; echo "b1 prio rc=$?"; tail -n 1 $D/bench_b1_prio.log | grep -o '"phase_ms[^}]*}'
