#!/usr/bin/env python
"""The batched persistent AR launch (valle_amd/csrc/persist_nb.hip, 2 .. 6 utterances per launch) on one GPU:
   python tools/persist_nb_probe.py --out gpurun_out/r6n [--batches 2 3 4] [--steps 300] [--tune] [--trace]
 1. microseconds per AR step of the launch chain, of the one-utterance launch and of the batched launch at BASELINE configs[1]'s shape;
 2. --tune: coordinate descent over the six first-sweep waits ("persist_naps", 4 bits per edge) per batch;
 3. --trace: the in-kernel timeline (option "persist_trace"): per stage, time computing and time waiting for the hand-off."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import valle_amd  # noqa: E402
from bench import P_PROMPT, S_TEXT, synth_inputs  # noqa: E402

STAGES = ["x", "qkv", "part", "att", "x2", "hid"]
EDGE_SHIFT = {"att": 0, "x": 4, "x2": 8, "hid": 12, "qkv": 16, "part": 20}


def decode(eng, X, Y, B, steps, opts):
    for k, v in opts.items():
        eng.set_option(k, v)
    eng.prefill(X[:B], [S_TEXT] * B, Y[:B], [P_PROMPT] * B)
    eng.generate(top_k=1, max_new=steps)
    tm = eng.timings()
    return tm["ar_ms"] * 1e3 / max(tm["ar_steps"], 1)


def timed(eng, X, Y, B, steps, opts, reps=2):
    decode(eng, X, Y, B, steps, opts)  # (re)captures the graph
    return min(decode(eng, X, Y, B, steps, opts) for _ in range(reps))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, nargs="*", default=[2, 3, 4, 5, 6])
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--out", default="gpurun_out/persist_nb_probe")
    ap.add_argument("--tune", action="store_true")
    ap.add_argument("--trace", action="store_true")
    ap.add_argument("--naps", type=lambda v: int(v, 0), default=-1)
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = valle_amd.VALLE(1024, 16, 12, prefix_mode=1, engine_dtype="bf16", max_batch=6).to(dev).eval()
    eng = model.engine_for(6, S_TEXT, P_PROMPT)
    eng.set_option("ignore_eos", 1)
    xs, ys = zip(*[synth_inputs(b) for b in range(6)])
    X, Y = torch.stack(xs).to(dev), torch.stack(ys).to(dev)
    report = {"shape": "d1024-L12-h16 bf16, S=%d, P=%d, %d AR steps, greedy" % (S_TEXT, P_PROMPT, args.steps), "rows": []}
    one = timed(eng, X, Y, 1, args.steps, {"persist": 1, "persist_naps": -1})
    report["one_utterance_us_per_step"] = round(one, 2)
    print("[one]", round(one, 2), flush=True)
    for B in args.batches:
        row = {"batch": B}
        row["chain_us"] = round(timed(eng, X, Y, B, args.steps, {"persist": 0}), 2)
        row["persist_us"] = round(timed(eng, X, Y, B, args.steps, {"persist": 1, "persist_naps": args.naps}), 2)
        assert eng.fetch_u32("persist_ran") == 1 and eng.fetch_u32("persist_fail") == 0
        row["tok_s_ar_only"] = {"chain": round(B * 8 / row["chain_us"] * 1e6), "persist": round(B * 8 / row["persist_us"] * 1e6)}
        print("[time]", json.dumps(row), flush=True)
        if args.tune:
            cur = 0x325756 if args.naps < 0 else args.naps
            best = timed(eng, X, Y, B, args.steps, {"persist": 1, "persist_naps": cur})
            hist = [[hex(cur), round(best, 2)]]
            for rnd in range(2):
                for edge in ("x", "qkv", "part", "att", "x2", "hid"):
                    sh = EDGE_SHIFT[edge]
                    v0 = (cur >> sh) & 15
                    for v in sorted(set(range(0, 16, 2)) | {min(15, v0 + 1), max(0, v0 - 1)}):
                        if v == v0:
                            continue
                        cand = (cur & ~(15 << sh)) | (v << sh)
                        t = timed(eng, X, Y, B, args.steps, {"persist": 1, "persist_naps": cand}, reps=2)
                        if t < best - 0.15:
                            best, cur = t, cand
                            hist.append([hex(cur), round(best, 2), edge])
                print("[tune]", B, rnd, hex(cur), round(best, 2), flush=True)
            row["tuned_naps"] = hex(cur)
            row["tuned_us"] = round(best, 2)
            row["tune_history"] = hist
        if args.trace:
            L = 12
            names = (["entry"] + ["L0." + n for n in STAGES[1:]] + [f"L{l}.{n}" for l in range(1, L) for n in STAGES] + ["final.x", "logits", "exit"])
            naps = int(row.get("tuned_naps", hex(args.naps) if args.naps >= 0 else "-1"), 0)
            for k, v in {"persist": 1, "persist_trace": 1, "persist_naps": naps}.items():
                eng.set_option(k, v)
            eng.prefill(X[:B], [S_TEXT] * B, Y[:B], [P_PROMPT] * B)
            eng.generate(top_k=1, max_new=64)
            raw = eng.fetch_persist_trace()
            eng.set_option("persist_trace", 0)
            n = len(names)
            tr = raw[:, :, : 3 * n].reshape(8, 256, n, 3)
            ok = (tr[..., 0] > 0).all(-1).all(-1)
            tr = tr[ok].double()
            t0, passes, t1 = tr[..., 0] / 100.0, tr[..., 1], tr[..., 2] / 100.0  # us
            wait = t1 - t0
            comp = torch.zeros_like(wait)
            comp[..., 1:] = t0[..., 1:] - t1[..., :-1]
            per = {}
            for i, nm in enumerate(names):
                if nm.startswith("L") and not nm.startswith("L0."):
                    per.setdefault(nm.split(".")[1], []).append((float(comp[..., i].mean()), float(wait[..., i].mean()), float(passes[..., i].mean())))
            summ = {st: {"compute_us": round(sum(v[0] for v in vs) / len(vs), 3), "wait_us": round(sum(v[1] for v in vs) / len(vs), 3),
                         "passes": round(sum(v[2] for v in vs) / len(vs), 2)} for st, vs in per.items()}
            i = names.index("logits")
            row["timeline"] = {"steps_seen": int(ok.sum()), "per_stage": summ,
                               "layer_us": round(sum(v["compute_us"] + v["wait_us"] for v in summ.values()), 3),
                               "final_x": {"compute_us": round(float(comp[..., names.index("final.x")].mean()), 3), "wait_us": round(float(wait[..., names.index("final.x")].mean()), 3)},
                               "logits": {"compute_us": round(float(comp[..., i].mean()), 3), "wait_us": round(float(wait[..., i].mean()), 3)},
                               "tail_us": round(float(comp[..., -1].mean()), 3)}
            print("[trace]", json.dumps(row["timeline"]), flush=True)
        report["rows"].append(row)
    eng.set_option("persist_naps", -1)
    with open(os.path.join(args.out, "persist_nb_probe.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report))


if __name__ == "__main__":
    main()
