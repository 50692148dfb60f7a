#!/usr/bin/env python
"""A/B timing of the batch-1 AR step under engine options (one process, interleaved rounds):
   python tools/ar_tune.py [--steps 200] [--rounds 3]
Prints microseconds per AR step (hipEvent time of the AR loop / steps) for each variant."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import valle_amd  # noqa: E402
from bench import P_PROMPT, S_TEXT, synth_inputs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--variants", nargs="*", default=None, help="e.g. qkv_attn=0 qa_nsplit=4,steps_per_graph=16")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = valle_amd.VALLE(1024, 16, 12, prefix_mode=1, engine_dtype=args.dtype, max_batch=args.batch).to(dev).eval()
    B = args.batch
    eng = model.engine_for(B, S_TEXT, P_PROMPT)
    X = torch.stack([synth_inputs(b)[0] for b in range(B)]).to(dev)
    Y = torch.stack([synth_inputs(b)[1] for b in range(B)]).to(dev)
    variants = [
        ("default", {}),
        ("unfused_qkv_attn", {"qkv_attn": 0}),
        ("fused_ns4", {"qa_nsplit": 4}),
        ("fused_ns16", {"qa_nsplit": 16}),
        ("spg16", {"steps_per_graph": 16}),
    ]
    if args.variants:
        variants = [("default", {})] + [(v, dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in v.split(","))) for v in args.variants]
    res = {name: [] for name, _ in variants}
    for r in range(args.rounds):
        for name, opts in variants:
            for k in ("nsplit", "gemv1_rpw", "gemv1_rpw_qkv", "gemv1_rpw_ffn1", "no_gemv1", "attn_nk", "steps_per_graph"):
                eng.set_option(k, 0)
            eng.set_option("qkv_attn", 1)
            eng.set_option("qa_nsplit", 8)
            eng.set_option("g1_shared", 1)
            eng.set_option("qa_waves", 4)
            eng.set_option("qa_qtemporal", 1)
            eng.set_option("qa_handoff", 1)
            eng.set_option("qa_nk", 4)
            for k, v in opts.items():
                eng.set_option(k, v)
            for rep in range(2):  # first pass (re)captures the graph
                eng.prefill(X, [S_TEXT] * B, Y, [P_PROMPT] * B)
                eng.generate(top_k=1, max_new=args.steps)
            tm = eng.timings()
            res[name].append(tm["ar_ms"] * 1e3 / max(tm["ar_steps"], 1))
    print(json.dumps({k: [round(x, 2) for x in v] for k, v in res.items()}))


if __name__ == "__main__":
    main()
