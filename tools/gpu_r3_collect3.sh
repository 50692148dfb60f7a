# round 3, third (last) collection after the two-fragment workgroups of the d = 1536 grids: whole GPU suite, smoke, default bench line.
cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
timeout 1500 python -m pytest tests -m gpu -q > $D/tests_all.log 2>&1; echo "all tests rc=$?"; tail -n 3 $D/tests_all.log
python -c "import __graft_entry__ as g; g.smoke()" > $D/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $D/smoke.log
timeout 900 python bench.py > $D/bench_default.log 2>&1; echo "default bench rc=$?"; tail -n 1 $D/bench_default.log | cut -c1-300
