# round 3: two W fragments per workgroup (d = 1536 grids) -- parity, then the configs[4] share with and without
cd $GRAFT_REPO_ROOT
D=gpurun_out/r3y; mkdir -p $D
timeout 600 python -m pytest tests/test_engine_gpu.py -q -k "compile_time_layout or fused_layernorm" 2>&1 | tail -6
for NFV in 1 0; do
  timeout 300 python bench.py --batch 32 --d-model 1536 --layers 24 --dtype fp8 --steps 2 --warmup 1 --cpu-frames 0 --no-side --opt gs_nf=$NFV > $D/bench_c5_nf$NFV.log 2>&1
  tail -n 1 $D/bench_c5_nf$NFV.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('gs_nf=$NFV', d['value'], d['ms_per_step'], d.get('phase_ms'), d['roofline'].get('launch_us'))"
done
