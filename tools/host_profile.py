#!/usr/bin/env python
"""Where the host time of one headline decode goes: cProfile over VALLE.inference_batch (BASELINE configs[1]) + the engine's own phase
times; prints wall per decode, the sum of the device phases and the top host functions.   python tools/host_profile.py [decodes]"""
import cProfile
import contextlib
import io
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import valle_amd  # noqa: E402
from bench import P_PROMPT, S_TEXT, synth_inputs  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = valle_amd.VALLE(1024, 16, 12, prefix_mode=1, engine_dtype="bf16").to(dev).eval()
eng = model.engine_for(1, S_TEXT, P_PROMPT)
eng.set_option("ignore_eos", 1)
x, y = synth_inputs(0)
X, Y = x[None].to(dev), y[None].to(dev)
lens = torch.tensor([S_TEXT], dtype=torch.int32)


def decode():
    with contextlib.redirect_stdout(io.StringIO()):
        return model.inference_batch(X, lens, Y, [P_PROMPT], None, top_k=1)


for _ in range(3):
    decode()
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
dev_ms = 0.0
for _ in range(n):
    decode()
    tm = eng.timings()
    dev_ms += tm["prefill_ms"] + tm["ar_ms"] + tm["nar_ms"]
torch.cuda.synchronize()
pr.disable()
wall = (time.perf_counter() - t0) * 1e3 / n
print(f"wall {wall:.3f} ms per decode, device phases {dev_ms / n:.3f} ms, other {wall - dev_ms / n:.3f} ms")
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18)
print(s.getvalue()[:6000])
