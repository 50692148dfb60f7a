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
    gl, out = bench.decode_step(m, X, [S] * B, Y, [P] * B, 1, 1, B, dev)  # through model.inference_batch(): the timed seam (B2)
    assert gl == [16 * S + 1] * B and len(out) == B
    gl_e, out_e = bench.decode_step(eng, X, [S] * B, Y, [P] * B, 1, 1, B, dev)  # the engine's own methods: the same decode
    assert gl_e == gl and all(torch.equal(a, b) for a, b in zip(out_e, out))
    for b in range(B):
        assert out[b].shape == (gl[b], 8) and out[b].device.type == "cuda"
        assert int(out[b].min()) >= 0 and int(out[b].max()) <= 1024
        assert torch.equal(out[b].to(torch.int16).to(torch.int64), out[b])
    again = vdist.gather_codes(out, B, 8, dev)
    assert all(torch.equal(a, b) for a, b in zip(again, out))
    elapsed = bench.timed_loop(lambda: bench.decode_step(m, X, [S] * B, Y, [P] * B, 1, 1, B, dev), 2, 1, 1, dev)
    assert elapsed > 0
    r = bench.rates(sum(gl) * 8 * 2, elapsed, 2 * B, gl[0])
    assert r["frames_per_s"] > 0 and abs(r["rtf"] * r["audio_s_per_wall_s"] - 1.0) < 1e-3


@pytest.mark.skipif(torch.cuda.device_count() >= 2, reason="needs a box with fewer than 2 GPUs")
def test_bench_gpus2_on_one_gpu_box_fails_loudly():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 2 and "needs 2 visible GPUs" in r.stderr and r.stdout.strip() == ""


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs (RCCL over xGMI)")
def test_bench_two_ranks_rccl_gathers_the_n1_codes():
    """Keeps the N > 1 path warm: `bench.py --gpus 2` spawns two RCCL ranks (one process per GPU), each decodes its own
    utterance, the gathered codes are the two single-GPU decodes in global order, and the line says so (world_size 2,
    backend nccl (RCCL)).  Skipped on 1-GPU boxes; the same code runs on 2 gloo ranks in tests/test_dist_cpu.py."""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    dump = os.path.join(ROOT, "gpurun_out", "bench2_codes.pt")
    os.makedirs(os.path.dirname(dump), exist_ok=True)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-side",
                        "--cpu-frames", "0", "--dump-codes", dump], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-800:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["world_size"] == 2 and line["config"]["backend"] == "nccl (RCCL)"
    assert line["scaling"] == "weak" and line["config"]["batch_per_gpu"] == 1
    got = torch.load(dump)
    assert len(got) == 2
    # the N = 1 decodes of the same two utterances, in this process
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    m = valle_amd.VALLE(1024, 16, 12, prefix_mode=1, engine_dtype="bf16").to(dev).eval()
    m.engine_for(1, bench.S_TEXT, bench.P_PROMPT).set_option("ignore_eos", 1)
    for i in range(2):
        x, y = bench.synth_inputs(i)
        _, out = bench.decode_step(m, x[None].to(dev), [bench.S_TEXT], y[None].to(dev), [bench.P_PROMPT], 1, 1, 1, dev)
        assert torch.equal(out[0].cpu(), got[i]), f"utterance {i}: the gathered codes differ from the single-GPU decode"
