"""N > 1 path on CPU: two processes, gloo backend (the GPU launch uses the same code over RCCL).
The decode path has no collective; the only exchange is the gather of the result codes."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_decode(i):
    """Deterministic stand-in for one utterance's (G_i, 8) code matrix (ragged lengths, ids < 1025)."""
    g = torch.Generator().manual_seed(100 + i)
    G = 5 + (i * 7) % 11
    return torch.randint(0, 1025, (G, 8), generator=g, dtype=torch.int64)


def _worker(rank, world, port, n_total, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import valle_amd  # noqa: F401
    from valle_amd import dist as vdist

    r, lr, w = vdist.init_process_group("gloo")
    assert (r, w) == (rank, world)
    seen = []

    def decode(lo, hi):
        seen.append((lo, hi))
        return [_fake_decode(i) for i in range(lo, hi)]

    out = vdist.decode_sharded(decode, n_total, 8, torch.device("cpu"))
    ok = len(out) == n_total and all(torch.equal(out[i], _fake_decode(i)) for i in range(n_total))
    q.put((rank, seen[0], ok))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("n_total", [2, 5, 8])
def test_batch_shard_and_gather_two_ranks(n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, ok0), (r1, s1, ok1) = res
    assert ok0 and ok1
    # contiguous, disjoint, complete split
    assert s0[0] == 0 and s0[1] == s1[0] and s1[1] == n_total


def test_shard_range_properties():
    sys.path.insert(0, ROOT)
    import valle_amd  # noqa: F401
    from valle_amd.dist import shard_range

    for n in (0, 1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


# ---- bench.py's own step / timing functions over 2 gloo ranks, with a stub in place of the HIP engine -------------------
class _StubEngine:
    """Engine-shaped stand-in (prefill / generate / nar / timings / ar_step_bytes): utterance b of a rank decodes to a
    code matrix that depends only on its INPUTS, so the gathered list can be checked against a single-process run."""

    def __init__(self):
        self.calls = 0

    def prefill(self, X, s_lens, Y, p_lens):
        self.X, self.Y = X, Y

    def generate(self, top_k=1, temperature=1.0, seed=0, allow_empty=False):
        self.gl = [4 + int(self.X[b].sum()) % 7 for b in range(self.X.shape[0])]
        return None, self.gl

    def nar(self, enroll):
        B, G = self.X.shape[0], max(self.gl)
        codes = torch.zeros(B, G, 8, dtype=torch.int64)
        for b in range(B):
            base = int(self.Y[b].sum()) % 1000
            codes[b] = (base + torch.arange(G)[:, None] * 8 + torch.arange(8)[None, :]) % 1025
        self.calls += 1
        return codes

    def timings(self):
        return dict(prefill_ms=1.0, ar_ms=2.0, nar_ms=3.0, ar_steps=max(self.gl))

    def ar_step_bytes(self, B, ctx):
        return 10 * B + ctx


def _stub_expected(i, S=6, P=5):
    import bench

    x, y = bench.synth_inputs(i, S, P)
    eng = _StubEngine()
    eng.prefill(x[None], [S], y[None], [P])
    _, gl = eng.generate()
    return eng.nar(None)[0, : gl[0]]


def _bench_worker(rank, world, port, B, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import bench
    from valle_amd import dist as vdist

    vdist.init_process_group("gloo")
    dev = torch.device("cpu")
    S, P = 6, 5
    X = torch.stack([bench.synth_inputs(rank * B + b, S, P)[0] for b in range(B)])
    Y = torch.stack([bench.synth_inputs(rank * B + b, S, P)[1] for b in range(B)])
    eng = _StubEngine()
    got = {}

    def step():
        return bench.decode_step(eng, X, [S] * B, Y, [P] * B, 1, world, world * B, dev)

    def on_step(r):
        got["gl"], got["out"] = r

    elapsed = bench.timed_loop(step, 3, 1, world, dev, on_step)
    ok = len(got["out"]) == world * B and all(torch.equal(got["out"][i], _stub_expected(i)) for i in range(world * B))
    q.put((rank, ok, eng.calls, elapsed))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_bench_step_and_timing_contract_two_ranks():
    """bench.py's decode_step (decode + gather) and timed_loop (warm-up, barrier-bracketed K steps, MAX over ranks) on 2 gloo
    ranks: every rank ends up with all utterances in global order, runs exactly W + K steps, and reports the same time."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, 3, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in res)
    assert [c for _, _, c, _ in res] == [4, 4]          # 1 warm-up + 3 timed steps on each rank
    assert res[0][3] == res[1][3] > 0                   # all_reduce(MAX): both ranks report the same elapsed time


def test_bench_refuses_more_gpus_than_visible():
    """`python bench.py --gpus 2` on a box with fewer GPUs must fail loudly (exit code 2, message), never run 1 rank and
    print n_gpus: 2.  (Here: 0 GPUs.)"""
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert "needs 2 visible GPUs" in r.stderr and r.stdout.strip() == ""
