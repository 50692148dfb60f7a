#!/usr/bin/env python
"""The persistent batch-1 AR step (option "persist", valle_amd/csrc/persist.hip) against the launch chain, on one GPU:
   python tools/persist_probe.py --out gpurun_out/r4a [--steps 300] [--rounds 2] [--check-steps 96]
 1. correctness: every logit of a teacher-free greedy decode must be BIT-IDENTICAL to the chain run with qa_nsplit = 16 and the
    same keys-per-lane split (persist_nk = qa_nk), for every prefetch depth; the give-up counter must stay 0;
 2. microseconds per AR step (hipEvent time of the AR loop / steps) of the chain and of every persistent variant, interleaved;
 3. the in-kernel timeline of the persistent step (option "persist_trace"): mean time between consecutive hand-offs."""
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


def reset(eng):
    for k in ("nsplit", "gemv1_rpw", "gemv1_rpw_qkv", "gemv1_rpw_ffn1", "no_gemv1", "attn_nk", "steps_per_graph"):
        eng.set_option(k, 0)
    for k, v in (("qkv_attn", 1), ("qa_nsplit", 8), ("g1_shared", 1), ("qa_waves", 4), ("qa_qtemporal", 1), ("qa_handoff", 1), ("qa_nk", 4),
                 ("persist", 0), ("persist_pf", 3), ("persist_nk", 2), ("persist_mode", 0x174), ("persist_naps", 0x325756), ("persist_sample", 1), ("act_bf16", 0), ("persist_trace", 0), ("trace_ar_logits", 0)):
        eng.set_option(k, v)


def decode(eng, X, Y, steps, opts, trace=False):
    reset(eng)
    for k, v in opts.items():
        eng.set_option(k, v)
    if trace:
        eng.set_option("trace_ar_logits", 1)
    eng.prefill(X, [S_TEXT], Y, [P_PROMPT])
    codes, gl = eng.generate(top_k=1, max_new=steps)
    out = {"codes": codes[0, : gl[0]].cpu(), "gl": gl[0]}
    if trace:
        out["logits"] = eng.fetch_ar_logits()[:, 0].clone()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--check-steps", type=int, default=96)
    ap.add_argument("--out", default="gpurun_out/persist_probe")
    ap.add_argument("--skip-check", action="store_true")
    ap.add_argument("--variants", nargs="*", default=["pf=3", "pf=3,sample=0", "pf=3,steps=8", "pf=3,mode=0x114", "pf=0", "pf=3,naps=0"],
                    help="persistent variants to time: comma-separated persist_* options, e.g. pf=0,mode=3,nk=2")
    ap.add_argument("--trace", nargs="*", default=["pf=3"])
    ap.add_argument("--dtype", default="bf16", help="engine mode of the probed model (bf16 | fp8w | fp32): the timing section only; --skip-check with the others")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = valle_amd.VALLE(1024, 16, 12, prefix_mode=1, engine_dtype=args.dtype).to(dev).eval()
    eng = model.engine_for(1, S_TEXT, P_PROMPT)
    eng.set_option("ignore_eos", 1)
    x, y = synth_inputs(0)
    X, Y = x[None].to(dev), y[None].to(dev)
    report = {}

    # ---- 1. bit-identity with the chain --------------------------------------------------------------------------------------
    if not args.skip_check:
        checks = []
        for nk, pf, md in ((2, 3, 0x174), (2, 0, 0x174), (2, 3, 0x134), (2, 0, 0x134), (2, 3, 0x114), (2, 0, 0x114), (2, 3, 0), (2, 3, 0x10)):
            if True:
                ab = (2 if md & 4 else 0) | (1 if md & 8 else 0)
                if md & 64:  # bf16 activation rows + v_dot2c: against the fp32-row form of the same launch (free-running: the record shows where they part)
                    ref = decode(eng, X, Y, args.check_steps, {"persist": 1, "persist_nk": nk, "persist_pf": pf, "persist_mode": md & ~64}, trace=True)
                elif md & 32:  # folded LayerNorm: against the three-barrier form of the same launch (fp32 re-association apart)
                    ref = decode(eng, X, Y, args.check_steps, {"persist": 1, "persist_nk": nk, "persist_pf": pf, "persist_mode": md & ~32}, trace=True)
                else:
                    ref = decode(eng, X, Y, args.check_steps, {"qa_nsplit": 16, "qa_nk": nk, "act_bf16": ab}, trace=True)
                got = decode(eng, X, Y, args.check_steps, {"persist": 1, "persist_nk": nk, "persist_pf": pf, "persist_mode": md, "qa_nsplit": 16, "qa_nk": nk}, trace=True)
                active = eng.fetch_u32("persist_active")
                fail = eng.fetch_u32("persist_fail")
                n = min(ref["logits"].shape[0], got["logits"].shape[0])
                diff = (ref["logits"][:n] - got["logits"][:n]).abs()
                neq = (ref["logits"][:n] != got["logits"][:n]).any(-1)
                first_bad = int(neq.nonzero()[0]) if neq.any() else -1
                rec = {"nk": nk, "pf": pf, "mode": md, "active": active, "fail": fail, "steps": n, "gl": [ref["gl"], got["gl"]],
                       "bit_identical": bool(not neq.any()) and ref["gl"] == got["gl"], "max_abs_diff": float(diff.max()),
                       "first_bad_step": first_bad, "tokens_equal": bool(torch.equal(ref["codes"], got["codes"])),
                       "logit_sigma": float(ref["logits"][:n].std())}
                if first_bad >= 0:
                    row = diff[first_bad]
                    rec["bad_rows_at_first"] = int((row > 0).sum())
                    rec["nan_at_first"] = int(torch.isnan(got["logits"][first_bad]).sum())
                checks.append(rec)
                print("[check]", json.dumps(rec), flush=True)
        report["checks"] = checks

    # ---- 2. timing -----------------------------------------------------------------------------------------------------------
    variants = [("chain_default", {})]
    for name in args.variants:
        opts = {"persist": 1}
        for kv in name.split(","):
            k, v = kv.split("=")
            opts[k if k == "steps_per_graph" else "persist_" + k] = int(v, 0)
        variants.append((name, opts))
    res = {name: [] for name, _ in variants}
    for r in range(args.rounds):
        for name, opts in variants:
            for rep in range(2):  # first pass (re)captures the graph
                out = decode(eng, X, Y, args.steps, opts)
            tm = eng.timings()
            res[name].append(round(tm["ar_ms"] * 1e3 / max(tm["ar_steps"], 1), 2))
        print("[time]", json.dumps(res), flush=True)
    report["us_per_step"] = res
    report["persist_fail_after_timing"] = eng.fetch_u32("persist_fail")

    # ---- 3. timeline (option "persist_trace"): thread 0 of every workgroup records, per hand-off, {wall clock when it began to wait,
    #         polling passes, wall clock when it had the data} -------------------------------------------------------------------
    L = 12
    for name in args.trace:
        opts = {"persist": 1, "persist_trace": 1}
        for kv in name.split(","):
            k, v = kv.split("=")
            opts[k if k == "steps_per_graph" else "persist_" + k] = int(v, 0)
        own = opts.get("persist_sample", 1) != 0  # the sampling step inside the launch: one more hand-off (the logits) per step
        names = (["entry"] + ["L0." + n for n in STAGES[1:]] + [f"L{l}.{n}" for l in range(1, L) for n in STAGES] + ["final.x"] +
                 (["logits"] if own else []) + ["exit"])
        reset(eng)
        for k, v in opts.items():
            eng.set_option(k, v)
        eng.prefill(X, [S_TEXT], Y, [P_PROMPT])
        eng.generate(top_k=1, max_new=64)
        raw = eng.fetch_persist_trace()
        n = len(names)
        tr = raw[:, :, : 3 * n].reshape(8, 256, n, 3)
        ok = (tr[..., 0] > 0).all(-1).all(-1)
        tr = tr[ok].double()
        t0, passes, t1 = tr[..., 0] / 100.0, tr[..., 1], tr[..., 2] / 100.0  # us
        wait = (t1 - t0)                       # (steps, wg, n)
        comp = torch.zeros_like(wait)
        comp[..., 1:] = t0[..., 1:] - t1[..., :-1]
        per = {}
        for i, nm in enumerate(names):
            if nm.startswith("L") and not nm.startswith("L0."):
                st = nm.split(".")[1]
                per.setdefault(st, []).append((float(comp[..., i].mean()), float(wait[..., i].mean()), float(passes[..., i].mean()),
                                               float(wait[..., i].amax(-1).mean())))
        summ = {st: {"compute_us": round(sum(v[0] for v in vs) / len(vs), 3), "wait_us": round(sum(v[1] for v in vs) / len(vs), 3),
                     "passes": round(sum(v[2] for v in vs) / len(vs), 2), "wait_us_slowest_wg": round(sum(v[3] for v in vs) / len(vs), 3)}
                for st, vs in per.items()}
        layer = sum(v["compute_us"] + v["wait_us"] for v in summ.values())
        body = (t1[..., -1].amax(-1) - t0[..., 0].amin(-1))
        rec = {"variant": name, "steps_seen": int(ok.sum()), "per_stage": summ, "layer_us": round(layer, 3), "kernel_body_us": round(float(body.mean()), 2),
               "entry_spread_us": round(float((t0[..., 0].amax(-1) - t0[..., 0].amin(-1)).mean()), 3),
               "final_x": {"compute_us": round(float(comp[..., names.index("final.x")].mean()), 3), "wait_us": round(float(wait[..., names.index("final.x")].mean()), 3)},
               "tail_us": round(float(comp[..., -1].mean()), 3)}
        if own:
            i = names.index("logits")
            rec["logits"] = {"compute_us": round(float(comp[..., i].mean()), 3), "wait_us": round(float(wait[..., i].mean()), 3), "passes": round(float(passes[..., i].mean()), 2)}
        if False:  # finer stamps: per layer 5 (out-proj stage) + 5 (linear2 stage)
            half = raw.shape[-1] // 2
            sb = raw[:, :, half: half + 10 * L].reshape(8, 256, L, 10)[ok].double() / 100.0
            lab = ["lds_read", "dot_reduce", "publish", "next_weights_issue"]
            o = sb[..., 1:5] - sb[..., 0:4]
            f = sb[..., 6:10] - sb[..., 5:9]
            rec["sub_outproj_us"] = {k: round(float(o[..., i].mean()), 3) for i, k in enumerate(lab)}
            rec["sub_linear2_us"] = {k: round(float(f[..., i].mean()), 3) for i, k in enumerate(lab)}
            ia = torch.tensor([names.index(f"L{l}.att") for l in range(L)])
            ih = torch.tensor([names.index(f"L{l}.hid") for l in range(L)])
            rec["sub_outproj_barrier_to_first"] = round(float((sb[..., 0] - t1[..., ia]).mean()), 3)
            rec["sub_linear2_barrier_to_first"] = round(float((sb[..., 5] - t1[..., ih]).mean()), 3)
        print("[trace]", json.dumps(rec), flush=True)
        report.setdefault("trace", []).append(rec)
        import csv
        with open(os.path.join(args.out, "persist_timeline_%s.csv" % name.replace("=", "").replace(",", "_")), "w", newline="") as f:
            wr = csv.writer(f)
            wr.writerow(["handoff", "compute_us_before", "wait_us_mean", "wait_us_max_over_wg", "passes_mean"])
            for i, nm in enumerate(names):
                wr.writerow([nm, round(float(comp[..., i].mean()), 3), round(float(wait[..., i].mean()), 3), round(float(wait[..., i].amax(-1).mean()), 3),
                             round(float(passes[..., i].mean()), 2)])
    with open(os.path.join(args.out, "persist_probe.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report))


if __name__ == "__main__":
    main()
