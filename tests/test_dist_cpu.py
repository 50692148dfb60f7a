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
