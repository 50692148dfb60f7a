#!/usr/bin/env python
"""Soak test of the batched persistent AR launch (csrc/persist_nb.hip):  python tools/persist_nb_stress.py SECONDS [STEPS]
Eight utterances of ragged lengths are decoded once each on the one-utterance launch (the references), then over and over in random
groups of 1 .. 6 in random order -- dense calls (vle_ar_generate) under changing first-sweep timings and launch lengths, and slot
sessions (vle_slots_*) with admissions while other slots are mid-decode.  Every AR token and every logit of every utterance must equal
its reference bit for bit whatever shared the launch, whatever ran before on the same granule buffers, and no wave may give up."""
import os
import random
import sys
import time

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import valle_amd  # noqa: E402
from valle_amd import ContinuousBatcher, Request  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
rng = random.Random(1)
L = 12
model = valle_amd.VALLE(1024, 16, L, prefix_mode=1, engine_dtype="bf16", max_batch=6).to(dev).eval()
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 96
N = 8
g = torch.Generator().manual_seed(3)
S = [int(torch.randint(8, 14, (1,), generator=g)) for _ in range(N)]
P = [int(torch.randint(4, 8, (1,), generator=g)) for _ in range(N)]  # (6 prefills packed stay below 128 rows: the same GEMM family as alone)
Xs, Ys = [], []
for i in range(N):
    x = torch.randint(3, 100, (S[i],), generator=g)
    x[0], x[-1] = 1, 2
    Xs.append(x)
    Ys.append(torch.randint(0, 1024, (P[i], 8), generator=g))
eng = model.engine_for(6, max(S), max(P))
eng.set_option("ignore_eos", 1)


def pack(ids):
    X = torch.zeros(len(ids), max(S[i] for i in ids), dtype=torch.int64)
    Y = torch.zeros(len(ids), max(P[i] for i in ids), 8, dtype=torch.int64)
    for j, i in enumerate(ids):
        X[j, : S[i]] = Xs[i]
        Y[j, : P[i]] = Ys[i]
    return X.to(dev), Y.to(dev)


def dense(ids, naps=-1, psteps=32):
    for k, v in (("persist", 1), ("persist_naps", naps), ("persist_steps", psteps), ("trace_ar_logits", 1)):
        eng.set_option(k, v)
    X, Y = pack(ids)
    eng.prefill(X, [S[i] for i in ids], Y, [P[i] for i in ids])
    codes, gl = eng.generate(top_k=1, max_new=steps)
    assert eng.fetch_u32("persist_ran") == 1, "the persistent launch did not run"
    lg = eng.fetch_ar_logits().clone()
    return [codes[j, : gl[j]].cpu() for j in range(len(ids))], lg, eng.fetch_u32("persist_fail")


ref_codes, ref_lg = {}, {}
for i in range(N):
    c, lg, fail = dense([i])
    assert fail == 0
    ref_codes[i], ref_lg[i] = c[0], lg[:, 0]
NAPS = [-1, 0, 0xFFFFFF, 0x0F0F0F, 0x123456, 0xF000F0, 0x654321]
t0 = time.time()
n = bad = sessions = 0
while time.time() - t0 < float(sys.argv[1]):
    if n % 5 == 4:  # a slot session over all eight requests on 2 .. 6 of the engine's slots
        nslots = rng.choice([2, 3, 4, 5, 6])
        order = list(range(N))
        rng.shuffle(order)
        cb = ContinuousBatcher(model, 6, max_text=max(S), max_prompt=max(P), steps_per_round=rng.choice([3, 8, 16]), harvest_min=1)
        cb.max_batch = nslots  # (uses nslots of the engine's six slots: the others stay stopped utterances of the launch)
        eng.set_option("persist_naps", rng.choice(NAPS))
        out = cb.decode([Request(Xs[i], Ys[i]) for i in order], top_k=1)
        ok = eng.fetch_u32("persist_ran") == 1 and eng.fetch_u32("persist_fail") == 0
        for j, i in enumerate(order):
            k = min(out[j].shape[0], ref_codes[i].numel())
            ok = ok and k == ref_codes[i].numel() and torch.equal(out[j][:k, 0].cpu(), ref_codes[i][:k])
        sessions += 1
        if not ok:
            bad += 1
            print("MISMATCH slot session", n, nslots, order, flush=True)
    else:
        ids = rng.sample(range(N), rng.choice([1, 2, 3, 4, 5, 6]))
        naps, psteps = rng.choice(NAPS), rng.choice([32, 32, 7, 1])
        codes, lg, fail = dense(ids, naps, psteps)
        ok = fail == 0
        for j, i in enumerate(ids):
            k = ref_codes[i].numel()
            ok = ok and torch.equal(codes[j], ref_codes[i]) and torch.equal(lg[1:k, j], ref_lg[i][1:k])
        if not ok:
            bad += 1
            print("MISMATCH dense", n, ids, hex(naps & 0xFFFFFF), psteps, fail, flush=True)
    n += 1
    if n % 50 == 0:
        print("iter", n, flush=True)
print("done", n, "rounds (", sessions, "slot sessions ) x", steps, "steps;", bad, "mismatches")
sys.exit(1 if bad else 0)
