#!/usr/bin/env python
"""TFLOP/s of the prefill / NAR GEMM (gemm_glds.hip) at the row counts of BASELINE configs[1] (1025 rows) and
configs[2] (65 600 rows), per tile-policy knob.   python tools/gemm_bench.py"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from valle_amd import ops  # noqa: E402

DEV = torch.device("cuda", 0)


def tflops(M, N, K, epi, reps=20):
    a = (torch.randn(M, K, device=DEV)).to(torch.bfloat16)
    w = (torch.randn(N, K, device=DEV) / math.sqrt(K)).to(torch.bfloat16)
    bias = torch.randn(N, device=DEV)
    r = torch.zeros(M, N, device=DEV) if epi == ops.EPI_RESID else None
    for _ in range(3):
        ops.linear(a, w, bias, epi, resid=r, ksplit=None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.linear(a, w, bias, epi, resid=r, ksplit=None)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * M * N * K * reps / (e0.elapsed_time(e1) * 1e-3) / 1e12


def main():
    shapes = [("qkv", 3072, 1024, ops.EPI_STORE), ("oproj", 1024, 1024, ops.EPI_RESID), ("ffn1", 4096, 1024, ops.EPI_RELU),
              ("ffn2", 1024, 4096, ops.EPI_RESID)]
    quick = "--quick" in sys.argv
    if "--persist" in sys.argv:  # round 6: the persistent tile loop of gemm_8ph.hip against the one-tile kernel, alternating
        for M in (65600, 32800, 16400):
            for rep in range(2):
                for v in (0, 7):
                    ops.tune("g8_persist", v)
                    row = [f"{name} {tflops(M, N, K, epi):7.1f}" for name, N, K, epi in shapes]
                    print(f"M={M:6d} g8_persist={v} TF/s: " + "  ".join(row), flush=True)
        ops.tune("g8_persist", 1)
        return
    for M in ((65600,) if quick else (65600, 8200, 1025)):
        for knobs in (({}, {"g8_nt": 1}, {"g8_nt": 2}, {"g8_nt": 3}, {"g8_dbg": 1}, {}) if quick else
                      ({}, {"g8_nt": 1}, {"g8_nt": 2}, {"g8_nt": 3}, {"g8_dbg": 4}, {"g8_dbg": 1}, {"g8_colgroup": 4}, {"glds_8ph": 0}, {"glds_8ph": 0, "glds_epi": 0}, {})):
            for k, v in knobs.items():
                ops.tune(k, v)
            row = [f"{name} {tflops(M, N, K, epi):7.1f}" for name, N, K, epi in shapes]
            print(f"M={M:6d} {str(knobs):22s} TF/s: " + "  ".join(row), flush=True)
            for k in knobs:
                ops.tune(k, {"glds_swz": 0, "glds_prio": 0, "glds_w8": 1, "glds_big": -1, "glds_8ph": -1, "g8_stagger": 1, "g8_colgroup": 0, "g8_dbg": 0, "glds_epi": 1, "g8_nt": 0}[k])


if __name__ == "__main__":
    main()
