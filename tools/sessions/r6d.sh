# round 6, session d: persistent tile loop, one vs two counted waits per K-tile: bit identity, GEMM TF/s, NAR phase at full C3 size
cd $GRAFT_REPO_ROOT
D=gpurun_out/r6d; mkdir -p $D
export TMPDIR=/tmp
timeout 300 python tools/fresh_box_probe.py --out $D/first > $D/probe.log 2>&1; echo "probe rc=$?"
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "persistent_tile_loop" > $D/tests_ops.log 2>&1; echo "ops tests rc=$?"; tail -n 3 $D/tests_ops.log
timeout 600 python tools/gemm_bench.py --persist > $D/gemm_persist.log 2>&1; echo "gemm bench rc=$?"; grep "M=" $D/gemm_persist.log
timeout 1500 python tools/nar_ab.py --batch 64 --reps 3 --steps 0 --opt g8_persist=0 --opt g8_persist=1 --opt g8_persist=9 --opt g8_persist=15 > $D/nar_ab_b64_full.log 2>&1; echo "nar_ab rc=$?"; tail -n 1 $D/nar_ab_b64_full.log
