#!/usr/bin/env python
"""Experiment: B utterances decoded as C independent sub-batches ("chains"), each with its own engine
and HIP stream, driven from its own host thread -- do the launch gaps of one chain fill with the
other chains' kernels?   python tools/chains_bench.py --batch 64 --chains 1 2 4"""
import argparse
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
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--chains", type=int, nargs="+", default=[1, 2, 4])
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--opt", action="append", default=[])
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    base = valle_amd.VALLE(1024, 16, 12, prefix_mode=1, engine_dtype="bf16")
    sd = base.state_dict()
    for C in args.chains:
        Bc = args.batch // C
        models, engs, inputs, streams = [], [], [], []
        for c in range(C):
            m = valle_amd.VALLE(1024, 16, 12, prefix_mode=1, engine_dtype="bf16", max_batch=Bc)
            m.load_state_dict(sd)
            m = m.to(dev).eval()
            e = m.engine_for(Bc, S_TEXT, P_PROMPT)
            e.set_option("ignore_eos", 1)
            for kv in args.opt:
                k, v = kv.split("=")
                e.set_option(k, int(v))
            X = torch.stack([synth_inputs(c * Bc + b)[0] for b in range(Bc)]).to(dev)
            Y = torch.stack([synth_inputs(c * Bc + b)[1] for b in range(Bc)]).to(dev)
            models.append(m); engs.append(e); inputs.append((X, Y)); streams.append(torch.cuda.Stream(dev))

        def run(c, out):
            with torch.cuda.stream(streams[c]):
                e = engs[c]
                X, Y = inputs[c]
                e.prefill(X, [S_TEXT] * Bc, Y, [P_PROMPT] * Bc)
                _, gl = e.generate(top_k=1, allow_empty=True)
                codes = e.nar(None)
                streams[c].synchronize()
                out[c] = (sum(gl) * 8, e.timings())

        best = None
        for rep in range(args.reps + 1):
            out = [None] * C
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ths = [threading.Thread(target=run, args=(c, out)) for c in range(C)]
            [t.start() for t in ths]
            [t.join() for t in ths]
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if rep > 0 and (best is None or dt < best[0]):
                best = (dt, sum(o[0] for o in out), [round(o[1]["ar_ms"], 1) for o in out], [round(o[1]["nar_ms"], 1) for o in out])
        print(f"batch {args.batch} chains {C} (x{Bc}): {best[1] / best[0]:.0f} tok/s  wall {best[0] * 1e3:.1f} ms  ar_ms {best[2]} nar_ms {best[3]}", flush=True)
        del models, engs, inputs, streams
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
