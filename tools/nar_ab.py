#!/usr/bin/env python
"""A/B of the prefill / NAR passes: engine options given on the command line, e.g.
    python tools/nar_ab.py --batch 1 --reps 5 --opt ln_fold=0 --opt ln_fold=1
prints the prefill / NAR milliseconds of BASELINE configs[1] (batch 1) or configs[2] (batch 64) decodes for every option set."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import valle_amd  # noqa: E402
from bench import P_PROMPT, S_TEXT, synth_inputs  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--steps", type=int, default=0, help="AR steps (0 = to the length cap: the NAR rows of the benchmark)")
ap.add_argument("--opt", action="append", default=[], help="name=value[,name=value]: one option set per flag")
args = ap.parse_args()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
B = args.batch
model = valle_amd.VALLE(1024, 16, 12, prefix_mode=1, engine_dtype="bf16", max_batch=B).to(dev).eval()
eng = model.engine_for(B, S_TEXT, P_PROMPT)
eng.set_option("ignore_eos", 1)
X = torch.stack([synth_inputs(b)[0] for b in range(B)]).to(dev)
Y = torch.stack([synth_inputs(b)[1] for b in range(B)]).to(dev)
res = {}
for rep in range(args.reps):
    for spec in args.opt or ["ln_fold=1"]:
        for kv in spec.split(","):
            k, v = kv.split("=")
            eng.set_option(k, int(v, 0))
        eng.prefill(X, [S_TEXT] * B, Y, [P_PROMPT] * B)
        eng.generate(top_k=1, max_new=args.steps)
        eng.nar(None)
        tm = eng.timings()
        res.setdefault(spec, []).append((round(tm["prefill_ms"], 3), round(tm["nar_ms"], 3)))
print(json.dumps({"batch": B, "prefill_nar_ms": res}))
