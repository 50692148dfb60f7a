"""Per-wave distribution of the in-kernel stamps of the batched AR step's GEMMs (option "ktrace"): is the time to a kernel's
LAST first-MFMA (profiles/*ktrace_b64_timeline.csv: ~5 us) every wave's time, or a few stragglers?  Prints, per kernel of one
layer, percentiles over the waves of: start - first start, mark1 - own start, end - own start."""
import argparse, ctypes as C, json, os, sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import valle_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--out", default="gpurun_out/ktrace_dist.json")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B = args.batch
    torch.manual_seed(0)
    model = valle_amd.VALLE(1024, 16, 12, prefix_mode=1, engine_dtype="bf16", max_batch=B).to(dev).eval()
    eng = model.engine_for(B, bench.S_TEXT, bench.P_PROMPT)
    eng.set_option("ignore_eos", 1)
    for kv in args.opt:
        eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    X = torch.stack([bench.synth_inputs(b)[0] for b in range(B)]).to(dev)
    Y = torch.stack([bench.synth_inputs(b)[1] for b in range(B)]).to(dev)

    def run(n):
        eng.prefill(X, [bench.S_TEXT] * B, Y, [bench.P_PROMPT] * B)
        eng.generate(top_k=1, max_new=n, allow_empty=True)

    run(28)
    eng.set_option("ktrace", 1)
    run(28)
    raw = torch.empty(32, 64, 2048, 4, dtype=torch.int64)
    n = eng.lib.vle_debug_fetch(eng.h, b"ktrace", C.c_void_p(raw.data_ptr()), raw.numel() * 8)
    assert n >= 0
    out = {}
    names = ["qkv", "attn", "out_proj", "ffn1", "ffn2"]
    for k in (5, 7, 8, 9):  # layer 1
        rows = []
        for s in range(10, 22):
            r = raw[s, k]
            v = r[r[:, 0] != -1].double() * 0.01
            if v.numel() == 0:
                continue
            t0 = v[:, 0].min()
            rows.append(torch.stack([v[:, 0] - t0, v[:, 1] - v[:, 0], v[:, 3] - v[:, 0], v[:, 1] - t0], dim=1))
        a = torch.cat(rows)
        q = torch.tensor([0.0, 0.1, 0.5, 0.9, 0.99, 1.0], dtype=torch.float64)
        out[names[k % 5]] = {
            "waves_per_launch": int(rows[0].shape[0]),
            "pct": [0, 10, 50, 90, 99, 100],
            "start_after_first_us": [round(float(x), 2) for x in torch.quantile(a[:, 0], q)],
            "mark1_after_own_start_us": [round(float(x), 2) for x in torch.quantile(a[:, 1], q)],
            "end_after_own_start_us": [round(float(x), 2) for x in torch.quantile(a[:, 2], q)],
            "mark1_after_first_start_us": [round(float(x), 2) for x in torch.quantile(a[:, 3], q)],
        }
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
