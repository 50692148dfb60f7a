cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
run() { tag=$1; shift; timeout 400 python bench.py "$@" > $D/bench_$tag.log 2>&1; echo "$tag rc=$?"; tail -n 1 $D/bench_$tag.log | grep -o '"value": [0-9.]*\|"phase_ms[^}]*}\|"frac": [0-9.]*' | head -3 | tr '\n' ' '; echo; }
run c1_fp32 --d-model 256 --layers 6 --nhead 4 --dtype fp32 --steps 3 --warmup 1 --cpu-frames 0 --no-c3
run c1_bf16 --d-model 256 --layers 6 --nhead 4 --dtype bf16 --steps 3 --warmup 1 --cpu-frames 0 --no-c3
run c2_fp32 --dtype fp32 --steps 2 --warmup 1 --cpu-frames 0 --no-c3
run c2_sampled --top-k -100 --steps 3 --warmup 1 --cpu-frames 0 --no-c3
run c2_b16 --batch 16 --steps 2 --warmup 1 --cpu-frames 0
run c2_b32 --batch 32 --steps 2 --warmup 1 --cpu-frames 0
