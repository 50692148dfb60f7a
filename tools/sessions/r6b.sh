# round 6, session b: the persistent tile loop of gemm_8ph.hip -- bit identity, GEMM TF/s A/B, NAR phase A/B at 64 utterances
cd $GRAFT_REPO_ROOT
D=gpurun_out/r6b; mkdir -p $D
export TMPDIR=/tmp
timeout 300 python tools/fresh_box_probe.py --out $D/first > $D/probe.log 2>&1; echo "probe rc=$?"
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "persistent_tile_loop or layernorm_folded or leftover_rows or tile_policy" > $D/tests_ops.log 2>&1; echo "ops tests rc=$?"; tail -n 5 $D/tests_ops.log
timeout 600 python tools/gemm_bench.py --persist > $D/gemm_persist.log 2>&1; echo "gemm bench rc=$?"; cat $D/gemm_persist.log
timeout 900 python tools/nar_ab.py --batch 64 --reps 3 --steps 8 --opt g8_persist=0 --opt g8_persist=1 > $D/nar_ab_b64.log 2>&1; echo "nar_ab rc=$?"; tail -n 1 $D/nar_ab_b64.log
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q > $D/tests_engine.log 2>&1; echo "engine tests rc=$?"; tail -n 3 $D/tests_engine.log
