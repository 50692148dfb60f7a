#!/usr/bin/env python
"""Soak test of the persistent batch-1 AR step's hand-offs:  python tools/persist_stress.py SECONDS [STEPS]
Decodes the same utterance over and over with the shipped persist_mode under every request schedule and a set of hand-off timings
(persist_naps); every logit of every decode must equal the first decode's, bit for bit, and no wave may give up waiting."""
import os
import sys
import time

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import valle_amd  # noqa: E402
from bench import P_PROMPT, S_TEXT, synth_inputs  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = valle_amd.VALLE(1024, 16, 12, prefix_mode=1, engine_dtype="bf16").to(dev).eval()
eng = model.engine_for(1, S_TEXT, P_PROMPT)
eng.set_option("ignore_eos", 1)
x, y = synth_inputs(0)
X, Y = x[None].to(dev), y[None].to(dev)
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 120
D2, FOLDED = 0x174, 0x134  # the shipped default (bf16 rows + v_dot2c: request schedules 0 and 3 only) and the fp32-row folded form (request schedules 0 and 3: the others are no longer compiled)
variants = [(D2, 3, 0x325756), (D2, 0, 0x325756), (D2, 3, 0), (D2, 3, 0xFFFFFF), (FOLDED, 3, 0x0F0F0F), (FOLDED, 0, 0x335854), (D2, 3, 0x123456), (D2, 0, 0x0F0F0F),
            (FOLDED, 3, 0x335854), (FOLDED, 0, 0)]
ref = {}
t0 = time.time()
n = bad = 0
while time.time() - t0 < float(sys.argv[1]):
    mode, pf, naps = variants[n % len(variants)]
    for k, v in (("persist", 1), ("persist_pf", pf), ("persist_mode", mode), ("persist_naps", naps), ("trace_ar_logits", 1)):
        eng.set_option(k, v)
    eng.prefill(X, [S_TEXT], Y, [P_PROMPT])
    codes, gl = eng.generate(top_k=1, max_new=steps)
    lg = eng.fetch_ar_logits()[:, 0].clone()
    fail = eng.fetch_u32("persist_fail")
    if mode not in ref:
        ref[mode] = lg
    if not torch.equal(ref[mode], lg) or fail or eng.fetch_u32("persist_ran") != 1:
        bad += 1
        print("MISMATCH", n, hex(mode), pf, hex(naps), fail, (ref[mode] - lg).abs().max().item(), flush=True)
    n += 1
    if n % 100 == 0:
        print("iter", n, flush=True)
print("done", n, "decodes x", steps, "steps;", bad, "mismatches")
sys.exit(1 if bad else 0)
