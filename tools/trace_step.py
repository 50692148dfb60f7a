"""Timestamped account of the hipGraph-replayed AR step from a rocprofv3 --kernel-trace CSV (SURVEY.md 8d "kernel-gap
timeline"):   python tools/trace_step.py <kernel_trace.csv> <out_prefix> [--steps 8] [--layers 12]

Finds the steady-state AR loop (the run of `ar_sample_kernel` launches), takes `--steps` consecutive steps from its
middle, and writes
  <out_prefix>_timeline.csv : every kernel of those steps -- start offset (us), duration, gap to the previous kernel's end
  <out_prefix>_summary.json : per kernel family  n, mean body us, mean gap-before us;  per step: sum of bodies, sum of gaps,
                              wall (first start -> next step's first start)
so that every microsecond of a step is attributed to a kernel body or to a boundary."""
import csv
import json
import sys
from collections import defaultdict


def short(name):
    n = name.split("(")[0]
    for key in ("gemv1_kernel", "decode_attn", "ar_sample", "gemm_skinny", "layernorm", "persist"):
        if key in n:
            # keep template args that distinguish the GEMV roles (PRO / EPI), drop the rest
            return n[n.index(key):][:90]
    return n[-60:]


def main():
    path, out = sys.argv[1], sys.argv[2]
    nsteps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 8
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    samp = [i for i, r in enumerate(rows) if "ar_sample" in r[2]]
    if len(samp) < nsteps + 4:
        print("not enough AR steps in the trace", len(samp))
        return 1
    mid = len(samp) // 2
    lo, hi = samp[mid] + 1, samp[mid + nsteps] + 1  # kernels after sample[mid] up to and including sample[mid + nsteps]
    win = rows[lo:hi]
    t0 = win[0][0]
    prev_end = rows[lo - 1][1]
    fam = defaultdict(lambda: [0, 0.0, 0.0])
    with open(out + "_timeline.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["idx", "start_us", "dur_us", "gap_before_us", "kernel"])
        for i, (s, e, n) in enumerate(win):
            gap = (s - prev_end) / 1e3
            w.writerow([i, round((s - t0) / 1e3, 3), round((e - s) / 1e3, 3), round(gap, 3), short(n)])
            k = short(n)
            fam[k][0] += 1
            fam[k][1] += (e - s) / 1e3
            fam[k][2] += gap
            prev_end = e
    steps = []
    for j in range(nsteps):
        a, b = samp[mid + j] + 1, samp[mid + j + 1] + 1
        seg = rows[a:b]
        body = sum(e - s for s, e, _ in seg) / 1e3
        wall = (rows[b][0] - seg[0][0]) / 1e3 if b < len(rows) else (seg[-1][1] - seg[0][0]) / 1e3
        steps.append(dict(kernels=len(seg), body_us=round(body, 2), wall_us=round(wall, 2), gaps_us=round(wall - body, 2)))
    summ = dict(
        window=f"{nsteps} consecutive AR steps from the middle of the loop ({len(samp)} sampled steps in the trace)",
        per_step=steps,
        mean_step=dict(body_us=round(sum(s["body_us"] for s in steps) / nsteps, 2), wall_us=round(sum(s["wall_us"] for s in steps) / nsteps, 2),
                       gaps_us=round(sum(s["gaps_us"] for s in steps) / nsteps, 2), kernels=steps[0]["kernels"]),
        families={k: dict(n=v[0], mean_body_us=round(v[1] / v[0], 3), mean_gap_before_us=round(v[2] / v[0], 3)) for k, v in sorted(fam.items())},
    )
    json.dump(summ, open(out + "_summary.json", "w"), indent=1)
    print(json.dumps(summ["mean_step"]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
