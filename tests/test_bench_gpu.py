"""GPU: bench.py's step function on the real engine (world size 1) and its refusal to fake a multi-GPU run."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402
import valle_amd  # noqa: E402
from valle_amd import dist as vdist  # noqa: E402


def test_decode_step_and_gather_on_real_engine_output_world1():
    """bench.decode_step -> (lengths, list of (G, 8) matrices) from the HIP engine; gather_codes at world size 1 is the
    identity on real device tensors; the int16 wire format round-trips every id the engine can emit."""
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    B, S, P = 3, 6, 10
    m = valle_amd.VALLE(128, 2, 2, prefix_mode=1, engine_dtype="bf16", max_batch=B).to(dev).eval()
    eng = m.engine_for(B, S, P)
    eng.set_option("ignore_eos", 1)
    X = torch.stack([bench.synth_inputs(b, S, P)[0] for b in range(B)]).to(dev)
    Y = torch.stack([bench.synth_inputs(b, S, P)[1] for b in range(B)]).to(dev)
    gl, out = bench.decode_step(eng, X, [S] * B, Y, [P] * B, 1, 1, B, dev)
    assert gl == [16 * S + 1] * B and len(out) == B
    for b in range(B):
        assert out[b].shape == (gl[b], 8) and out[b].device.type == "cuda"
        assert int(out[b].min()) >= 0 and int(out[b].max()) <= 1024
        assert torch.equal(out[b].to(torch.int16).to(torch.int64), out[b])
    again = vdist.gather_codes(out, B, 8, dev)
    assert all(torch.equal(a, b) for a, b in zip(again, out))
    elapsed = bench.timed_loop(lambda: bench.decode_step(eng, X, [S] * B, Y, [P] * B, 1, 1, B, dev), 2, 1, 1, dev)
    assert elapsed > 0


@pytest.mark.skipif(torch.cuda.device_count() >= 2, reason="needs a box with fewer than 2 GPUs")
def test_bench_gpus2_on_one_gpu_box_fails_loudly():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 2 and "needs 2 visible GPUs" in r.stderr and r.stdout.strip() == ""
