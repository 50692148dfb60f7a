#!/usr/bin/env python
"""Experiment (round 6): 64 utterances as TWO independent sub-batches of 32, each on its own engine, host thread and HIP stream --
with the two streams confined to DISJOINT halves of the CUs (hipExtStreamCreateWithCUMask).  Round 4 measured the same split on
ordinary streams (tools/chains_bench.py: +1.4 %: the kernels of two queues interleave, they do not overlap); disjoint CU sets force the
HBM-bound attention of one half to run beside the latency-bound weight GEMMs of the other.
   python tools/cumask_probe.py [--batch 64] [--modes none halves interleaved]"""
import argparse
import ctypes as C
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import valle_amd  # noqa: E402
from bench import P_PROMPT, S_TEXT, synth_inputs  # noqa: E402

hip = C.CDLL("libamdhip64.so")


def masked_stream(dev, bits):
    """a HIP stream whose kernels run on the CUs named in `bits` (list of CU indices), wrapped for torch"""
    words = [0] * 8
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    arr = (C.c_uint32 * 8)(*words)
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, arr)
    assert rc == 0, f"hipExtStreamCreateWithCUMask -> {rc}"
    return torch.cuda.ExternalStream(st.value, device=dev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--modes", nargs="+", default=["one", "none", "halves", "interleaved"])
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--no-nar", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    base = valle_amd.VALLE(1024, 16, 12, prefix_mode=1, engine_dtype="bf16")
    sd = base.state_dict()
    for mode in args.modes:
        Cn = 1 if mode == "one" else 2
        Bc = args.batch // Cn
        if mode == "halves":
            masks = [list(range(0, 128)), list(range(128, 256))]
        elif mode == "interleaved":
            masks = [list(range(0, 256, 2)), list(range(1, 256, 2))]
        else:
            masks = [None] * Cn
        models, engs, inputs, streams = [], [], [], []
        for c in range(Cn):
            m = valle_amd.VALLE(1024, 16, 12, prefix_mode=1, engine_dtype="bf16", max_batch=Bc)
            m.load_state_dict(sd)
            m = m.to(dev).eval()
            e = m.engine_for(Bc, S_TEXT, P_PROMPT)
            e.set_option("ignore_eos", 1)
            X = torch.stack([synth_inputs(c * Bc + b)[0] for b in range(Bc)]).to(dev)
            Y = torch.stack([synth_inputs(c * Bc + b)[1] for b in range(Bc)]).to(dev)
            models.append(m); engs.append(e); inputs.append((X, Y))
            streams.append(torch.cuda.Stream(dev) if masks[c] is None else masked_stream(dev, masks[c]))

        def run(c, out):
            with torch.cuda.stream(streams[c]):
                e = engs[c]
                X, Y = inputs[c]
                e.prefill(X, [S_TEXT] * Bc, Y, [P_PROMPT] * Bc)
                _, gl = e.generate(top_k=1, allow_empty=True)
                if not args.no_nar:
                    e.nar(None)
                streams[c].synchronize()
                out[c] = (sum(gl) * 8, e.timings())

        best = None
        for rep in range(args.reps + 1):
            out = [None] * Cn
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ths = [threading.Thread(target=run, args=(c, out)) for c in range(Cn)]
            [t.start() for t in ths]
            [t.join() for t in ths]
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if rep > 0 and (best is None or dt < best[0]):
                best = (dt, sum(o[0] for o in out), [round(o[1]["ar_ms"], 1) for o in out], [round(o[1]["nar_ms"], 1) for o in out])
        print(f"[cumask] batch {args.batch} mode {mode} ({Cn} x {Bc}): {best[1] / best[0]:.0f} tok/s  wall {best[0] * 1e3:.1f} ms  ar_ms {best[2]} nar_ms {best[3]}", flush=True)
        del models, engs, inputs, streams
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
