# round 6, session c: persistent tile loop at the full C3 size (64 utterances to the length cap), per epilogue class
cd $GRAFT_REPO_ROOT
D=gpurun_out/r6c; mkdir -p $D
export TMPDIR=/tmp
timeout 300 python tools/fresh_box_probe.py --out $D/first > $D/probe.log 2>&1; echo "probe rc=$?"
timeout 1500 python tools/nar_ab.py --batch 64 --reps 3 --steps 0 --opt g8_persist=0 --opt g8_persist=1 --opt g8_persist=3 --opt g8_persist=7 > $D/nar_ab_b64_full.log 2>&1; echo "nar_ab rc=$?"; tail -n 1 $D/nar_ab_b64_full.log
timeout 600 python tools/nar_ab.py --batch 8 --reps 3 --steps 0 --opt g8_persist=0 --opt g8_persist=7 > $D/nar_ab_b8_full.log 2>&1; echo "nar_ab b8 rc=$?"; tail -n 1 $D/nar_ab_b8_full.log
