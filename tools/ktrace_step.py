"""Un-perturbed timeline of the hipGraph-replayed batch-1 AR step (SURVEY.md 8d "kernel-gap timeline"), taken by the
kernels themselves (engine option "ktrace": every wave stamps the 100 MHz wall clock at entry and before its epilogue),
plus the steps-per-graph sweep that prices a graph-replay boundary.

    python tools/ktrace_step.py [--out gpurun_out/ktrace] [--dtype bf16]

Writes <out>_timeline.csv (one row per kernel of the mean step: gap before it, start ramp, body, end spread) and
<out>_summary.json (sums per kernel family, step wall, steps-per-graph sweep)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import valle_amd  # noqa: E402

NAMES5 = ["qkv(LN1+in_proj+KV write)", "decode_attention", "out_proj(merge+resid)", "ffn1(LN2+relu)", "ffn2(+resid)"]
# batch 1 with the fused launch (engine option qkv_attn = 1, the default): 4 launches per layer
NAMES4 = ["qkv+attention(LN1+in_proj+KV write+old keys)", "out_proj(merge+own key+resid)", "ffn1(LN2+relu)", "ffn2(+resid)"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/ktrace")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--spg", default="1,2,4,8,16,32,64")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--d-model", type=int, default=1024)
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--kpl", type=int, default=0, help="launches per layer (0 = 4 at batch 1 unless --opt qkv_attn=0, else 5)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    L, B = args.layers, args.batch
    model = valle_amd.VALLE(args.d_model, 16, L, prefix_mode=1, engine_dtype=args.dtype, max_batch=B).to(dev).eval()
    eng = model.engine_for(B, bench.S_TEXT, bench.P_PROMPT)
    eng.set_option("ignore_eos", 1)
    for kv in args.opt:
        eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    X = torch.stack([bench.synth_inputs(b)[0] for b in range(B)]).to(dev)
    Y = torch.stack([bench.synth_inputs(b)[1] for b in range(B)]).to(dev)

    def run(max_new=0):
        eng.prefill(X, [bench.S_TEXT] * B, Y, [bench.P_PROMPT] * B)
        eng.generate(top_k=1, max_new=max_new, allow_empty=True)
        return eng.timings()

    run(); run()
    base = run()
    out = {"ar_us_per_step_default": round(base["ar_ms"] * 1e3 / base["ar_steps"], 2)}
    # ---- steps-per-graph sweep: t_step = t0 + X / spg  => X = cost of one graph-replay boundary
    sweep = {}
    for spg in [int(v) for v in args.spg.split(",")]:
        eng.set_option("steps_per_graph", spg)
        run()
        best = min(run()["ar_ms"] for _ in range(3))
        sweep[spg] = round(best * 1e3 / base["ar_steps"], 2)
    out["us_per_step_by_steps_per_graph"] = sweep
    eng.set_option("steps_per_graph", 8)
    # ---- in-kernel stamps, 28 steps (the 32 slots never wrap), context ~300
    eng.set_option("ktrace", 1)
    run(28)
    eng.set_option("ktrace", 1)  # re-arm (clears the buffer)
    tm = run(28)
    kt = eng.fetch_ktrace().double() * 0.01  # us
    KPL = args.kpl or (4 if (B == 1 and "qkv_attn=0" not in args.opt and args.d_model // 16 in (64, 128)) else 5)
    NAMES = NAMES4 if KPL == 4 else NAMES5
    nk = KPL * L + 2
    assert nk <= 64 or B == 1, 'the trace holds 64 kernels per step'
    nk = min(nk, 64)
    steps = list(range(9, 25))  # two whole 8-step graph replays, away from both ends
    rows = []
    fam = {}
    for k in range(nk):
        gap = ramp = body = spread = ph1 = ph2 = ph1f = 0.0
        for s in steps:
            cur = kt[s, k]
            prev_end = kt[s, k - 1, 7] if k > 0 else kt[s - 1, nk - 1, 7]
            gap += float(cur[0] - prev_end)
            ramp += float(cur[1] - cur[0])
            body += float(cur[7] - cur[0])
            spread += float(cur[7] - cur[6])
            ph1 += float(cur[3] - cur[0])
            ph1f += float(cur[2] - cur[0])
            ph2 += float(cur[5] - cur[3])
        n = len(steps)
        name = NAMES[k % KPL] if k < KPL * L else ("final_LN+predict" if k == KPL * L else "sample+stop+embed")
        rows.append((k, name, gap / n, ramp / n, body / n, spread / n, ph1 / n, ph2 / n, ph1f / n))
        f = fam.setdefault(name, [0, 0.0, 0.0, 0.0])
        f[0] += 1; f[1] += gap / n; f[2] += body / n; f[3] += ramp / n
    wall = sum(float(kt[s + 1, 0, 0] - kt[s, 0, 0]) for s in steps) / len(steps)
    # graph boundary: gap before kernel 0 of the first step of each replay vs inside a replay
    g0 = [float(kt[s, 0, 0] - kt[s - 1, nk - 1, 7]) for s in range(3, 28)]
    out.update(
        ktrace_note="stamps: first/last wave entry, first/last wave before its epilogue store; 100 MHz clock (10 ns); "
                    "body = last end - first start, gap = first start - previous kernel's last end (store drain + boundary + dispatch)",
        ktrace_ar_us_per_step=round(tm["ar_ms"] * 1e3 / tm["ar_steps"], 2),
        mean_step=dict(wall_us=round(wall, 2), sum_body_us=round(sum(r[4] for r in rows), 2), sum_gap_us=round(sum(r[2] for r in rows), 2),
                       kernels=nk),
        gap_before_first_kernel_of_each_step_us=[round(v, 2) for v in g0],
        families={k: dict(n=v[0], sum_gap_us=round(v[1], 2), sum_body_us=round(v[2], 2), mean_gap_us=round(v[1] / v[0], 3),
                          mean_body_us=round(v[2] / v[0], 3), mean_start_ramp_us=round(v[3] / v[0], 3)) for k, v in fam.items()},
    )
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out + "_timeline.csv", "w") as f:
        f.write("idx,kernel,gap_before_us,start_ramp_us,body_us,end_spread_us,start_to_last_mark1_us,mark1_to_last_mark2_us,start_to_first_mark1_us\n")
        for r in rows:
            f.write(f"{r[0]},{r[1]},{r[2]:.3f},{r[3]:.3f},{r[4]:.3f},{r[5]:.3f},{r[6]:.3f},{r[7]:.3f},{r[8]:.3f}\n")
    json.dump(out, open(args.out + "_summary.json", "w"), indent=1)
    print(json.dumps(out)[:3000])


if __name__ == "__main__":
    main()
