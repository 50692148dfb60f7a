#!/usr/bin/env python
"""Per-launch cost of the batch-2..64 AR-step GEMMs (gemm_skinny.hip) as dependent graph chains, weights
rotated over > 256 MiB so every launch streams W from HBM:  v1 (X fragments -> VGPR) vs v2 (X staged in LDS,
wn = 1 | 2), split-K slice counts, and LayerNorm as its own launch vs fused in the v2 prologue.
    python tools/gs_bench.py [--M 64]"""
import argparse
import json
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from valle_amd import ops  # noqa: E402
from tools.op_chain_bench import chain_us  # noqa: E402

DEV = torch.device("cuda", 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=64)
    args = ap.parse_args()
    M = args.M
    torch.manual_seed(0)
    out = {}
    shapes = [("qkv 3072x1024", 3072, 1024, ops.EPI_STORE, True), ("ffn1 4096x1024", 4096, 1024, ops.EPI_RELU, True),
              ("oproj 1024x1024", 1024, 1024, ops.EPI_RESID, False), ("ffn2 1024x4096", 1024, 4096, ops.EPI_RESID, False),
              ("logits 1025x1024", 1025, 1024, ops.EPI_F32, True)]
    for name, N, K, epi, has_ln in shapes:
        nW = max(8, int(300e6 / (N * K * 2)))
        Ws = [(torch.randn(N, K, device=DEV) / math.sqrt(K)).to(torch.bfloat16) for _ in range(nW)]
        bias = torch.randn(N, device=DEV) * 0.1
        x32 = torch.randn(M, K, device=DEV)
        xb = x32.to(torch.bfloat16)
        g, b = torch.ones(K, device=DEV), torch.zeros(K, device=DEV)
        r = torch.zeros(M, N, device=DEV)
        res = {}
        for variant, wn in ((1, 0), (2, 1), (2, 2)):
            ops.tune("gs_variant", variant)
            ops.tune("gs_wn", wn)
            for ks in ((1, 2, 4, 8) if K >= 2048 else (1, 2)):
                if K % (256 * ks):
                    continue
                try:
                    fn = lambda i: ops.linear(xb, Ws[i % nW], bias, epi, resid=r if epi == ops.EPI_RESID else None, ksplit=ks)
                    res[f"v{variant} wn{wn} ks{ks}"] = round(chain_us(fn), 2)
                except Exception as e:  # shape not covered by this variant
                    res[f"v{variant} wn{wn} ks{ks}"] = str(e)[:40]
        if has_ln:
            ops.tune("gs_variant", 0)
            for wn in (1, 2):
                ops.tune("gs_wn", wn)
                res[f"LN launch + v2 wn{wn}"] = round(chain_us(lambda i: ops.linear(ops.layernorm(x32, g, b, out_dtype=torch.bfloat16), Ws[i % nW], bias, epi, ksplit=1)), 2)
                res[f"LN fused v2 wn{wn}"] = round(chain_us(lambda i: ops.ln_linear(x32, g, b, Ws[i % nW], bias, epi, ksplit=1)), 2)
        ops.tune("gs_variant", 0)
        ops.tune("gs_wn", 0)
        out[f"M={M} {name}"] = res
        print(name, json.dumps(res), flush=True)
        del Ws
    x32 = torch.randn(M, 1024, device=DEV)
    g, b = torch.ones(1024, device=DEV), torch.zeros(1024, device=DEV)
    print("layernorm alone", round(chain_us(lambda i: ops.layernorm(x32, g, b, out_dtype=torch.bfloat16)), 2), flush=True)


if __name__ == "__main__":
    main()
