#!/usr/bin/env python
"""Ragged workload on one GPU: N requests with text lengths S ~ U[smin, 47] (so G = 16 S + 1 frames each, EOS ignored)
and 3 s prompts, C2 architecture bf16 -- static batches of `max_batch` in arrival order (each runs to its longest
member) vs continuous batching (valle_amd/serving.py).   python tools/serve_bench.py [--n 192] [--max-batch 64]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import valle_amd  # noqa: E402
from valle_amd import ContinuousBatcher, Request  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=192)
    ap.add_argument("--max-batch", type=int, default=64)
    ap.add_argument("--smin", type=int, default=8)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--harvest-min", type=int, nargs="+", default=[1, 8, 16, 32])
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    m = valle_amd.VALLE(1024, 16, 12, prefix_mode=1, engine_dtype=args.dtype, max_batch=args.max_batch).to(dev).eval()
    g = torch.Generator().manual_seed(7)
    S = torch.randint(args.smin, 48, (args.n,), generator=g).tolist()
    reqs = []
    for i in range(args.n):
        x = torch.randint(3, 100, (S[i],), generator=g)
        x[0], x[-1] = 1, 2
        reqs.append(Request(x.to(dev), torch.randint(0, 1024, (225, 8), generator=g).to(dev)))
    tokens = sum(16 * s + 1 for s in S) * 8
    B = args.max_batch
    eng = m.engine_for(B, 47, 225)
    eng.set_option("ignore_eos", 1)

    def static():
        for i0 in range(0, args.n, B):
            chunk = reqs[i0:i0 + B]
            X = torch.zeros(len(chunk), 47, dtype=torch.int64, device=dev)
            for j, r in enumerate(chunk):
                X[j, : r.text.numel()] = r.text
            Y = torch.stack([r.prompt for r in chunk])
            eng.prefill(X, [int(r.text.numel()) for r in chunk], Y, [225] * len(chunk))
            eng.generate(top_k=1, allow_empty=True)
            eng.nar(None)

    runs = [("static", static)]
    cbs = {}
    for h in args.harvest_min:
        cbs[h] = ContinuousBatcher(m, B, 47, 225, steps_per_round=8, harvest_min=h)
        runs.append((f"continuous harvest_min={h}", lambda h=h: cbs[h].decode(reqs, top_k=1)))
    for name, fn in runs:
        fn()  # warm-up (graph capture)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{name:28s}: {tokens / dt:10.0f} tok/s  ({dt * 1e3:.0f} ms for {args.n} requests, {tokens} tokens, max_batch {B})", flush=True)
    print("scheduler stats", {h: c.stats for h, c in cbs.items()})


if __name__ == "__main__":
    main()
