# round 2, call I: codec tests + timing
cd $GRAFT_REPO_ROOT
D=gpurun_out/$1; mkdir -p $D
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_codec_gpu.py -q -x > $D/tests_codec.log 2>&1; echo "codec tests rc=$?"; tail -n 15 $D/tests_codec.log
timeout 300 python - > $D/codec_time.log 2>&1 <<'PY'
import time, torch, sys
sys.path.insert(0, '.')
import valle_amd
from oracle import encodec_oracle as eo
sd = eo.make_state_dict(0)
dec = valle_amd.EncodecDecoder(sd, device="cuda:0")
codes = torch.randint(0, 1024, (753, 8)).cuda()
for _ in range(2): dec.decode_codes(codes)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): w = dec.decode_codes(codes)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print(f"codec decode of 753 frames (10.04 s of audio): {dt*1e3:.2f} ms  -> RTF {dt/10.04:.5f}")
PY
tail -n 2 $D/codec_time.log
