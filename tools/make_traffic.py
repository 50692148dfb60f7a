#!/usr/bin/env python
"""profiles/ar_step_traffic.json -- HBM bytes per batch-1 AR step from the rocprofv3 PMC passes (tools/gpu_r6_collect.sh: FETCH_SIZE and
WRITE_SIZE in SEPARATE runs over `bench.py --steps 1 --warmup 1 --cpu-frames 0 --no-graph --no-c3 --no-fp32`), keyed by the hash of
the step's kernel sources (bench.kernel_set_hash) so that bench.py stops reporting it once the kernels change.

    python tools/make_traffic.py gpurun_out/<tag>/pmc_FETCH_SIZE_by_kernel.csv gpurun_out/<tag>/pmc_WRITE_SIZE_by_kernel.csv [steps]

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts the 128-byte requests of wide coalesced streams at 64 bytes
(/opt/skills/guides/MI355X_MICROARCH.md, HBM): it is doubled.  `steps` = AR steps the profiled run made (default: the run's
753 + 753: one warm-up and one timed decode)."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

STEP_KERNELS = ("pstep_kernel", "gemv1s_kernel", "gemv1_kernel", "qkv_attn1_kernel", "decode_attn_kernel", "ar_sample_kernel", "skinny_kernel")


def total(path):
    tot, rows = 0.0, {}
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            if any(k in name for k in STEP_KERNELS):
                tot += float(r["Counter_Sum"])
                rows[name[:90]] = rows.get(name[:90], 0.0) + float(r["Counter_Sum"])
    return tot, rows


def main():
    fetch, frows = total(sys.argv[1])
    write, wrows = total(sys.argv[2])
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2 * 753
    per_step = (2.0 * fetch + write) * 1024.0 / steps
    out = {
        "kernel_set": bench.kernel_set_hash(),
        "bytes_per_step": int(per_step),
        "fetch_kib_x2_per_step": round(2.0 * fetch / steps, 1),
        "write_kib_per_step": round(write / steps, 1),
        "steps": steps,
        "source": f"{os.path.relpath(sys.argv[1], ROOT)} + {os.path.relpath(sys.argv[2], ROOT)} (rocprofv3 --pmc, separate passes; FETCH_SIZE x 2 per the gfx950 correction)",
        "note": "the AR step's kernels only (prefill / NAR launches excluded); algorithmic bytes at the mean context: 336.4 MB.  With the "
                "persistent step the counter also sees every sweep of the in-launch hand-offs (L1-bypassing loads, served by the "
                "memory-side cache / HBM: ~40 KB per workgroup and layer per successful sweep, more when a sweep is repeated)",
        "by_kernel_fetch_kib": {k: round(v, 1) for k, v in sorted(frows.items(), key=lambda kv: -kv[1])[:6]},
    }
    with open(os.path.join(ROOT, "profiles", "ar_step_traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
