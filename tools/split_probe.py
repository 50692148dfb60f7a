#!/usr/bin/env python
"""Batch-64 decode (BASELINE configs[2]/[3]) as ONE engine of 64 utterances against G independent engines of 64 / G utterances on
their own streams, driven from G host threads:  python tools/split_probe.py [--groups 1 2 4] [--steps 2]
The utterances of a batch never interact, so the groups are independent decodes; the question is whether the HBM-bound attention of
one group overlaps the latency-bound weight GEMMs of another.  Prints wall-clock tokens/s and checks the codes against G = 1."""
import argparse
import contextlib
import json
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import valle_amd  # noqa: E402
from bench import P_PROMPT, S_TEXT, synth_inputs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--groups", type=int, nargs="*", default=[1, 2, 4])
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--max-new", type=int, default=0)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    B = args.batch
    torch.manual_seed(0)
    base = valle_amd.VALLE(1024, 16, 12, prefix_mode=1, engine_dtype="bf16", max_batch=B)
    sd = base.state_dict()
    X = torch.stack([synth_inputs(b, S_TEXT)[0] for b in range(B)]).to(dev)
    Y = torch.stack([synth_inputs(b, S_TEXT)[1] for b in range(B)]).to(dev)
    ref = None
    for G in args.groups:
        n = B // G
        models = []
        for g in range(G):
            m = valle_amd.VALLE(1024, 16, 12, prefix_mode=1, engine_dtype="bf16", max_batch=n)
            m.load_state_dict(sd)
            m = m.to(dev).eval()
            m.engine_for(n, S_TEXT, P_PROMPT).set_option("ignore_eos", 1)
            models.append(m)
        outs = [None] * G

        def run(g):
            with torch.cuda.device(dev):
                xs, ys = X[g * n:(g + 1) * n], Y[g * n:(g + 1) * n]
                outs[g] = models[g].inference_batch(xs, torch.tensor([S_TEXT] * n, dtype=torch.int32), ys, [P_PROMPT] * n, None, top_k=1,
                                                    seed=0, max_new=args.max_new)

        def once():
            with contextlib.redirect_stdout(sys.stderr):
                if G == 1:
                    run(0)
                else:
                    th = [threading.Thread(target=run, args=(g,)) for g in range(G)]
                    for t in th:
                        t.start()
                    for t in th:
                        t.join()
            torch.cuda.synchronize(dev)

        once()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            once()
        dt = (time.perf_counter() - t0) / args.steps
        codes = [c for o in outs for c in o]
        tokens = sum(int(c.shape[0]) for c in codes) * 8
        same = None
        if ref is None:
            ref = [c.cpu() for c in codes]
        else:
            same = all(torch.equal(a, c.cpu()) for a, c in zip(ref, codes))
        tms = [m.engine_for(n, S_TEXT, P_PROMPT).timings() for m in models]
        print(json.dumps({"groups": G, "per_group": n, "s_per_decode": round(dt, 4), "tok_per_s": round(tokens / dt, 1), "codes_equal_g1": same,
                          "ar_ms": [round(t["ar_ms"], 1) for t in tms], "nar_ms": [round(t["nar_ms"], 1) for t in tms],
                          "prefill_ms": [round(t["prefill_ms"], 1) for t in tms]}), flush=True)
        del models


if __name__ == "__main__":
    main()
