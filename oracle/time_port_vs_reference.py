"""TEST INFRASTRUCTURE ONLY -- calibrates bench.py's `cpu_baseline` (kind "port") against the UNMODIFIED reference.

bench.py times the oracle (oracle/valle_oracle.py, the literal no-KV-cache restatement of valle/models/valle.py:961-1137) on
the GPU box's host cores, because /root/reference does not exist there.  This script, run in the BUILD container where the
reference tree is present, times both on the same short decode -- BASELINE configs[1]'s architecture (d1024-L12-h16, fp32), the
benchmark's 225-frame prompt, a 4-token text so that the reference's own `16 * S + 1` cap (valle.py:1047) ends the greedy decode
after 65 frames -- on the same threads, checks that they produce the same tokens, and writes the ratio to
profiles/cpu_port_vs_reference.json, which bench.py carries in `cpu_baseline.port_vs_reference`.

    python oracle/time_port_vs_reference.py [--threads 8] [--reps 2]
"""
import argparse
import contextlib
import io
import json
import os
import platform
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import ref_import  # noqa: E402
from oracle import valle_oracle as vo  # noqa: E402
from oracle.make_golden import build_reference  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--reps", type=int, default=2, help="timed repetitions of each (the minimum is reported, all are listed)")
    ap.add_argument("--text", type=int, default=4, help="text tokens S: the decode runs to the reference's cap of 16 S + 1 frames")
    ap.add_argument("--prompt", type=int, default=225)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "cpu_port_vs_reference.json"))
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    vm = ref_import.import_reference()
    cfg = vo.OracleConfig(d_model=1024, nhead=16, num_layers=12, prefix_mode=1)
    sd = vo.make_state_dict(cfg, 0)
    x, x_lens, y = vo.make_inputs(args.text, args.prompt, 1234)
    model = build_reference(vm, cfg, sd)

    def run_reference():
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            return model.inference(x, x_lens, y, enroll_x_lens=None, top_k=1, temperature=1.0)

    def run_port():
        return vo.inference(sd, cfg, x, x_lens, y, None, top_k=1, temperature=1.0, kv_cache=False)

    t_ref, t_port, out_ref, out_port = [], [], None, None
    for _ in range(args.reps):  # interleaved, so that a frequency / co-tenant drift hits both alike
        t0 = time.perf_counter()
        out_ref = run_reference()
        t_ref.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        out_port = run_port()
        t_port.append(time.perf_counter() - t0)
    same = out_ref.shape == out_port.shape and bool(torch.equal(out_ref, out_port))
    frames = int(out_ref.shape[1])
    tokens = frames * int(out_ref.shape[2])
    rec = {
        "what": "the unmodified reference (valle/models/valle.py::VALLE.inference, imported from /root/reference through oracle/ref_import.py) "
                "and the oracle port (oracle/valle_oracle.py::inference, kv_cache=False) on the same decode, same process, same threads",
        "config": {"d_model": 1024, "nhead": 16, "num_layers": 12, "dtype": "fp32", "text": args.text, "prompt": args.prompt, "frames": frames,
                   "top_k": 1, "threads": args.threads},
        "host": {"cpu": platform.processor() or platform.machine(), "cores_visible": os.cpu_count(), "torch": torch.__version__},
        "tokens_identical": same,
        "reference_s": [round(t, 3) for t in t_ref],
        "port_s": [round(t, 3) for t in t_port],
        "reference_tok_s": round(tokens / min(t_ref), 2),
        "port_tok_s": round(tokens / min(t_port), 2),
        "port_over_reference_speed": round(min(t_ref) / min(t_port), 4),
        "generator": "oracle/time_port_vs_reference.py",
    }
    with open(args.out, "w") as f:
        json.dump(rec, f, indent=1)
        f.write("\n")
    print(json.dumps(rec))
    assert same, "the port and the reference disagree on this decode"


if __name__ == "__main__":
    main()
