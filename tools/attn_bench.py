#!/usr/bin/env python
"""TFLOP/s of the prefill / NAR attention at the shapes of BASELINE configs[1] (one 1025-row sequence), configs[2] (64 of them) and
configs[4]'s share (32 sequences, dh 96), round 1's kernel (attn_v2 = 0) vs attn_mfma2.hip.   python tools/attn_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from valle_amd import ops  # noqa: E402

DEV = torch.device("cuda", 0)


def bench(B, N, nhead, dh, causal, reps=10):
    d = nhead * dh
    g = torch.Generator(device="cpu").manual_seed(0)
    qkv = torch.randn(B * N, 3 * d, generator=g).to(torch.bfloat16).to(DEV)
    so = torch.arange(0, (B + 1) * N, N, dtype=torch.int32, device=DEV)
    tl = torch.full((B,), 47 if causal else 0, dtype=torch.int32, device=DEV)
    for _ in range(2):
        ops.attention(qkv, so, tl, nhead, causal)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.attention(qkv, so, tl, nhead, causal)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flops = 4.0 * B * nhead * N * N * dh * (0.5 if causal else 1.0)
    return ms, flops / (ms * 1e-3) / 1e12


def main():
    if "--quick" in sys.argv:  # one shape, the default policy, few launches: the target of a rocprofv3 --pmc pass
        ms, tf = bench(64, 1025, 16, 64, False, reps=3)
        print(f"C3 NAR default policy: {ms:7.3f} ms {tf:6.1f} TF")
        return
    if "--lsum" in sys.argv:  # round 3: softmax row sums on the MFMA pipe (attn_lsum = 1, default) vs VALU adds
        for name, B, N, H, dh, causal in [("C2 NAR", 1, 1025, 16, 64, False), ("C3 NAR", 64, 1025, 16, 64, False), ("C3 prefill", 64, 272, 16, 64, True),
                                          ("C5 share NAR", 32, 1025, 16, 96, False)]:
            row = []
            for v in (1, 0, 1, 0):
                ops.tune("attn_lsum", v)
                ms, tf = bench(B, N, H, dh, causal)
                row.append(f"lsum={v}: {ms:7.3f} ms {tf:6.1f} TF")
            ops.tune("attn_lsum", 1)
            print(f"{name:14s} " + " | ".join(row), flush=True)
        return
    for name, B, N, H, dh, causal in [("C2 NAR", 1, 1025, 16, 64, False), ("C3 NAR", 64, 1025, 16, 64, False), ("C3 prefill", 64, 272, 16, 64, True),
                                      ("C5 share NAR", 32, 1025, 16, 96, False)]:
        row = []
        for tag, knobs in [("v1", dict(attn_v2=0)), ("v2 pad32 exact-max", dict(attn_v2=2, attn_mode=1, attn_defer=0)),
                           ("v2 pad32", dict(attn_v2=2, attn_mode=1)), ("v2 dma exact-max", dict(attn_v2=2, attn_mode=2, attn_defer=0)),
                           ("v2 dma", dict(attn_v2=2, attn_mode=2, attn_ring=2)), ("v2 dma tr-read V", dict(attn_v2=2, attn_mode=3)), ("v2 dma tr-read V q128", dict(attn_v2=2, attn_mode=3, attn_q128=1)), ("v2 pad32 q128", dict(attn_v2=2, attn_mode=1, attn_q128=1)),
                           ("v2 dma q128", dict(attn_v2=2, attn_mode=2, attn_q128=1)), ("default policy", dict(attn_v2=1))]:
            if tag != "default policy":
                ops.tune("attn_q128", 0)
            for k, v in knobs.items():
                ops.tune(k, v)
            ms, tf = bench(B, N, H, dh, causal)
            row.append(f"{tag}: {ms:7.3f} ms {tf:6.1f} TF")
            ops.tune("attn_v2", 1); ops.tune("attn_xcd", 1); ops.tune("attn_q128", -1); ops.tune("attn_mode", 3); ops.tune("attn_defer", 8); ops.tune("attn_ring", 0)
        print(f"{name:13s} B={B:3d} N={N} H={H} dh={dh} c={int(causal)}\n    " + "\n    ".join(row), flush=True)


if __name__ == "__main__":
    main()
