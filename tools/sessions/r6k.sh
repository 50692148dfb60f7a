cd $GRAFT_REPO_ROOT
D=gpurun_out/r6k; mkdir -p $D
export TMPDIR=/tmp
timeout 300 python tools/fresh_box_probe.py --out $D/first > $D/probe.log 2>&1; echo "probe rc=$?"
timeout 300 python tools/host_profile.py 20 > $D/host_profile.log 2>&1; echo "host profile rc=$?"; head -n 30 $D/host_profile.log | cut -c1-160
timeout 1200 python -m pytest tests/test_engine_gpu.py tests/test_serving_gpu.py tests/test_bench_gpu.py -x -q > $D/tests.log 2>&1; echo "tests rc=$?"; tail -n 3 $D/tests.log
