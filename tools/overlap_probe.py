#!/usr/bin/env python
"""Do the kernels of two HIP streams overlap on this chip, or interleave?  Reads a rocprofv3 --kernel-trace CSV of
`tools/cumask_probe.py --modes none --no-nar` (two engines of 32 utterances, each replaying its own AR-step graphs on its own stream)
and reports, for the AR-step kernels: sum of durations, union of busy time, time with >= 2 kernels in flight, and the mean duration per
kernel family -- beside the same figures of a one-stream run (--modes one).
   python tools/overlap_probe.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        n = r["Kernel_Name"]
        if not any(k in n for k in ("decode_attn_kernel", "gemm_skinny", "ar_sample_kernel")):
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("<")[0].split("::")[-1] + ("<" + n.split("<")[1][:12] if "<" in n else ""), r.get("Queue_Id", "?")))
rows.sort()
print("kernels:", len(rows), "queues:", sorted({q for *_, q in rows}))
ev = []
for s, e, _, _ in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = over = 0
depth, last = 0, ev[0][0]
for t, d in ev:
    if depth >= 1: busy += t - last
    if depth >= 2: over += t - last
    depth += d; last = t
tot = sum(e - s for s, e, _, _ in rows)
print(f"sum of durations {tot / 1e6:.1f} ms, union busy {busy / 1e6:.1f} ms, >= 2 in flight {over / 1e6:.1f} ms ({100.0 * over / busy:.1f} % of busy), span {(rows[-1][1] - rows[0][0]) / 1e6:.1f} ms")
fam = defaultdict(list)
for s, e, n, _ in rows:
    fam[n].append(e - s)
for n, v in sorted(fam.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print(f"  {n:40s} calls {len(v):6d}  mean {sum(v) / len(v) / 1e3:7.2f} us  median {v[len(v) // 2] / 1e3:7.2f} us  p90 {v[int(len(v) * 0.9)] / 1e3:7.2f} us")
