#!/usr/bin/env python
"""Per-kernel cost of the batch-1 AR-step kernels as dependent chains (torch CUDA-graph capture of
the stand-alone operators, 60 launches per replay, weights rotated over > 256 MiB so every launch
streams from HBM like a real decode step).   python tools/op_chain_bench.py"""
import json
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import valle_amd  # noqa: E402
from valle_amd import ops  # noqa: E402

DEV = torch.device("cuda", 0)
d, H, dh = 1024, 16, 64


def chain_us(fn, n=60, reps=20):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for i in range(3):
            fn(i)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(n):
            fn(i)
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (reps * n)


def main():
    torch.manual_seed(0)
    out = {}
    nW = 60
    for name, N, K, kind in [("qkv-like LN+store 3072x1024", 3072, 1024, "ln"), ("ffn1 LN+relu 4096x1024", 4096, 1024, "lnrelu"),
                             ("ffn2 plain+resid 1024x4096", 1024, 4096, "resid"), ("logits LN+store 1025x1024", 1025, 1024, "ln"),
                             ("oproj-like plain+resid 1024x1024", 1024, 1024, "resid")]:
        Ws = [(torch.randn(N, K, device=DEV) / math.sqrt(K)).to(torch.bfloat16) for _ in range(nW)]
        bias = torch.randn(N, device=DEV) * 0.1
        g, b = torch.ones(K, device=DEV), torch.zeros(K, device=DEV)
        x = torch.randn(1, K, device=DEV)
        r = torch.zeros(1, N, device=DEV)
        if kind == "ln":
            fn = lambda i: ops.linear_skinny(x, Ws[i % nW], None, 0, gamma=g, beta=b)
        elif kind == "lnrelu":
            fn = lambda i: ops.linear_skinny(x, Ws[i % nW], bias, 1, gamma=g, beta=b)
        else:
            fn = lambda i: ops.linear_skinny(x, Ws[i % nW], bias, 2, resid=r)
        out[name] = round(chain_us(fn), 2)
        del Ws
    # decode attention (+ merge kernel) and out-proj with fused merge, context 650 of 1026
    ctx_max = 1026
    kcs = [torch.randn(1, H, ctx_max, dh, device=DEV).to(torch.bfloat16) for _ in range(12)]
    vcs = [torch.randn(1, H, ctx_max, dh, device=DEV).to(torch.bfloat16) for _ in range(12)]
    q = torch.randn(1, d, device=DEV)
    kl = torch.tensor([650], dtype=torch.int32, device=DEV)
    for ns in (1, 2, 4, 8, 16):
        out[f"decode_attn nsplit={ns} (partials only)"] = round(chain_us(lambda i: ops.decode_attention(q, kcs[i % 12], vcs[i % 12], kl, nsplit=ns, merged=False)), 2)
    Wo = [(torch.randn(d, d, device=DEV) / 32).to(torch.bfloat16) for _ in range(nW)]
    bo = torch.zeros(d, device=DEV)
    r = torch.zeros(1, d, device=DEV)
    for ns in (1, 4, 8, 16):
        _, ws = ops.decode_attention(q, kcs[0], vcs[0], kl, nsplit=ns, merged=False)
        out[f"oproj attn-merge nsplit={ns}"] = round(chain_us(lambda i: ops.attn_out_proj(ws, Wo[i % nW], bo, r, H, ns)), 2)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
