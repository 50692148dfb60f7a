#!/usr/bin/env python
"""bench.py -- audio-tokens/s of the AR+NAR decode hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]

A "step" = one full pass of the hot path over one batch of synthetic utterances: prefill of
[text; 3 s prompt], the AR loop to the reference's length cap (G = 16*S + 1 = 753 frames = 10.04 s
at S = 47), and the 7 NAR stages.  Inputs are already resident in HBM when the timed region starts.
N > 1: one process per GPU (torch.distributed, RCCL), the batch is sharded (weak scaling, B
utterances per GPU); no collective on the decode path, one all_gather of the result codes per step.

Prints ONE JSON line on rank 0 (contract in the task statement) carrying `roofline` (AR step =
the HBM-bound dominant phase; algorithmic bytes of SURVEY.md 8d / measured hipEvent time) and
`cpu_baseline` (the CPU oracle -- a restatement of the reference's no-KV-cache algorithm -- timed
on the host cores on a bounded sample; N > 1 lines carry the committed N = 1 measurement with its provenance) and
`eager_gpu_baseline` (the same restatement as PyTorch-ROCm eager ops on the same GPU).

Scaling.  Per-GPU work is the SAME at every N ("scaling": "weak"): `value` is BASELINE configs[1] -- one utterance per GPU --
at N = 1, 2, 4, 8, and every line also carries `c3_batch64`: 64 utterances per GPU (configs[2] at N = 1, configs[3] at
N = 8), run by all ranks under the same barrier-bracketed timing, with its own AR (HBM) and NAR (MFMA) roofline fractions.
Scaling efficiency of either workload = its value at N / (N x its value at N = 1), from the lines alone (`scale_ref`).
Timed region = the reference's own seam (SURVEY.md 8d "wall-time from inference() entry"): every decode goes through
`model.inference_batch()` -- the method `VALLE.inference()` is a batch-1 wrapper of (valle/models/valle.py:961) -- with its
asserts, length conversions, engine lookup and EOS prints inside the number (prints redirected to stderr).
`batch4` (N = 1) is 4 utterances per GPU on the batched persistent launch (csrc/persist_nb.hip);
`sampled` (N = 1) is the headline workload at the reference's own default sampling (top_k = -100, temperature = 1.0,
valle/models/valle.py:967-968); `s200` (N = 1) a realistic text length (S = 200 -> 3201 frames = 42.7 s of audio).
`frames_per_s` / `rtf` (wall seconds per second of generated audio at 75 frames/s) accompany every tokens/s figure.
`fp32_exact` (N = 1) is the same decode in engine mode fp32 -- the mode whose greedy token ids are bit-identical to the
reference (tests/test_parity_sizes_gpu.py) -- timed in the same process; `c5_share_fp8w` (N = 1) is the per-GPU share of BASELINE
configs[4] (d1536-L24-h16, 32 utterances) in the config's weight format: fp8 (e4m3) weights, bf16 activations on the bf16 MFMA -- the
mode that holds the 5 % sigma parity bar; the experimental fp8-MFMA mode only with --c5-fp8.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
S_TEXT, P_PROMPT = 47, 225  # 47 phonemes, 3 s x 75 Hz prompt
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "ar_step_traffic.json")    # tools/make_traffic.py from the PMC passes
CPU_CACHE_FILE = os.path.join(ROOT, "profiles", "cpu_baseline_n1.json")  # the N = 1 line's cpu_baseline, committed
AR_STEP_KERNEL_SOURCES = ("persist.hip", "persist_dev.h", "gemv1_dev.h", "sampling_dev.h", "gemv1.hip", "sampling.hip", "common.h")  # the batch-1 AR step's kernels
N1_REF_FILE = os.path.join(ROOT, "profiles", "bench_n1_reference.json")  # value / c3_batch64.value of the committed N = 1 line (scale_ref)
FRAME_RATE = 75.0  # EnCodec frames per second of audio (valle/data/tokenizer.py: 24 kHz / 320)


def _code_only(src: str) -> str:
    """C++ source without comments and blank space: what the compiler sees (string literals in these files contain no `//`)."""
    import re

    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    return "\n".join(" ".join(line.split()) for line in src.splitlines() if line.strip())


def kernel_set_hash() -> str:
    """sha256[:16] over the CODE (comments and white space stripped) of the sources of the kernels one batch-1 AR step launches:
    `roofline.traffic` (HBM bytes per step from the PMC counters) is only reported while the kernels are the ones it was measured on."""
    import hashlib

    h = hashlib.sha256()
    for name in AR_STEP_KERNEL_SOURCES:
        with open(os.path.join(ROOT, "valle_amd", "csrc", name), "r", encoding="utf-8") as f:
            h.update(_code_only(f.read()).encode())
    return h.hexdigest()[:16]


def measured_traffic(args, B):
    """HBM bytes per AR step of the default workload (C2, batch 1, bf16) from profiles/ar_step_traffic.json -- rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE in separate passes, FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md -- or
    (None, reason) when the file is missing, was measured on other kernels, or the workload is not the one it describes."""
    if not (B == 1 and args.dtype == "bf16" and args.d_model == 1024 and args.layers == 12 and not args.opt):
        return None, "measured for the default workload only (C2, batch 1, bf16)"
    try:
        with open(TRAFFIC_FILE) as f:
            t = json.load(f)
    except OSError:
        return None, "profiles/ar_step_traffic.json missing"
    if t.get("kernel_set") != kernel_set_hash():
        return None, f"stale: measured on kernel set {t.get('kernel_set')}, the kernels have changed since (re-run tools/gpu_r6_collect.sh + tools/make_traffic.py)"
    return int(t["bytes_per_step"]), t.get("source", "profiles/")


def synth_inputs(index: int, S: int = S_TEXT, P: int = P_PROMPT):
    """ids uniform in [3,100) with BOS=1 / EOS=2, codes uniform in [0,1024); seed 1234+index."""
    g = torch.Generator().manual_seed(1234 + index)
    x = torch.randint(3, 100, (S,), generator=g, dtype=torch.int64)
    x[0], x[-1] = 1, 2
    y = torch.randint(0, 1024, (P, 8), generator=g, dtype=torch.int64)
    return x, y


def cpu_baseline(sd_cpu, d_model, nhead, num_layers, frames: int):
    """The CPU oracle (literal no-KV-cache restatement of valle/models/valle.py:961-1137), fp32, on
    the host cores, on a bounded sample: the first `frames` frames of utterance 0 (+ the 7 NAR
    stages over them).  Only this leg of bench.py touches oracle/.  The sample's rate is NOT the full-length rate: the literal
    algorithm re-runs all S+P+t rows every step and its attention grows quadratically, so the whole 753-frame utterance is far
    slower per token -- `reference_full` carries the unmodified reference's measured full-length figure, and `port_vs_reference` the
    calibration of this port against the unmodified reference on the same short decode (oracle/time_port_vs_reference.py, run in the
    build container where /root/reference exists; profiles/cpu_port_vs_reference.json)."""
    from oracle import valle_oracle as vo

    # intra-op threads actually used: the reference's per-step ops are small (one utterance), more
    # than ~16 threads only adds synchronisation cost (256 threads ran this sample 50x slower)
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    cfg = vo.OracleConfig(d_model=d_model, nhead=nhead, num_layers=num_layers, prefix_mode=1)
    x, y = synth_inputs(0)
    xl = torch.tensor([S_TEXT], dtype=torch.int32)
    t0 = time.perf_counter()
    yy = vo.ar_decode(sd_cpu, cfg, x[None], xl, y[None], top_k=1, temperature=1.0, kv_cache=False, max_new=frames)
    t_ar = time.perf_counter() - t0
    codes = vo.nar_decode(sd_cpu, cfg, x, yy[0], y, P_PROMPT)
    dt = time.perf_counter() - t0
    t_nar = dt - t_ar
    n_frames = codes.shape[1]
    n_tok = n_frames * codes.shape[2]
    G_full = 16 * S_TEXT + 1
    ctx0 = S_TEXT + P_PROMPT
    try:
        with open(os.path.join(ROOT, "profiles", "cpu_port_vs_reference.json")) as f:
            cal = json.load(f)
        pvr = dict(port_over_reference_speed=cal["port_over_reference_speed"], tokens_identical=cal["tokens_identical"],
                   reference_tok_s=cal["reference_tok_s"], port_tok_s=cal["port_tok_s"], config=cal["config"],
                   source="profiles/cpu_port_vs_reference.json (oracle/time_port_vs_reference.py: the unmodified reference and this port timed "
                          "on the same decode in the build container -- the reference tree does not travel to the GPU box)")
    except Exception as err:  # noqa: BLE001
        pvr = {"error": repr(err)[:120]}
    return dict(
        value=round(n_tok / dt, 3), unit="audio-tokens/s", cores=cores, kind="port",
        sample=f"first {n_frames} of {G_full} frames of utterance 0 (ctx {ctx0}..{ctx0 + frames}) "
               f"+ 7 NAR stages, fp32, {dt:.1f} s (AR {t_ar:.1f} s, NAR {t_nar:.1f} s)",
        reference_full=dict(
            value=9.8, unit="audio-tokens/s", cores=8, kind="reference", seconds=614.7, frames_per_s=1.22, rtf=61.0,
            source="BASELINE.md section 2: the UNMODIFIED valle/models/valle.py::VALLE.inference at this workload's full length "
                   "(S=47, P=225 -> 753 frames), fp32, greedy, 8 threads of the survey container's Xeon (the reference tree does not "
                   "travel to the GPU box, so it cannot be re-timed there; the port above is what runs on this box's cores)"),
        port_vs_reference=pvr,
    )


def eager_gpu_baseline(sd_cpu, d_model, nhead, num_layers, frames, dev):
    """The same literal (no KV cache) restatement of valle.py:961-1137, fp32, as plain PyTorch-ROCm eager ops on the
    SAME MI355X -- what running the reference with model.to("cuda") amounts to ("a fairer secondary
    baseline").  Bounded sample like cpu_baseline; only this leg and cpu_baseline touch oracle/."""
    from oracle import valle_oracle as vo

    cfg = vo.OracleConfig(d_model=d_model, nhead=nhead, num_layers=num_layers, prefix_mode=1)
    sd = {k: v.to(dev) for k, v in sd_cpu.items()}
    x, y = synth_inputs(0)
    xs, xl, ys = x[None].to(dev), torch.tensor([S_TEXT], dtype=torch.int32), y[None].to(dev)
    vo.inference(sd, cfg, xs, xl, ys, None, top_k=1, kv_cache=False, max_new=4)  # warm-up (rocBLAS / MIOpen handles)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    codes = vo.inference(sd, cfg, xs, xl, ys, None, top_k=1, kv_cache=False, max_new=frames)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    n_tok = codes.shape[1] * codes.shape[2]
    return dict(value=round(n_tok / dt, 3), unit="audio-tokens/s", kind="port (PyTorch-ROCm eager ops on the same GPU, fp32, no KV cache)",
                sample=f"first {codes.shape[1]} of 753 frames of utterance 0 + 7 NAR stages, {dt:.2f} s")


MFMA_PEAK_TFS = 2500.0  # dense bf16 MFMA peak (/opt/skills/guides/MI355X_MICROARCH.md)


def side_leg(sd, args, dev, rank, world, B, dtype, steps=2, warmup=1, label="", S=S_TEXT, top_k=None, model=None):
    """A second workload measured beside the headline under the SAME rules (inputs resident, whole decode calls through
    model.inference_batch() incl. the gather, barrier + device sync on both sides, MAX over ranks, tokens summed over ranks):
    `c3_batch64` = BASELINE configs[2] / [3] (64 utterances per GPU), `fp32_exact` = the headline workload in the token-exact
    engine mode, `sampled` / `s200` = the headline model at the reference's default sampling / at a realistic text length.
    An extra object of the JSON line, never `value`.  N > 1: the ranks agree that every one of them built its model before any
    of them enters a collective (a rank that failed would leave the others waiting in the barrier for ever)."""
    import valle_amd

    top_k = args.top_k if top_k is None else top_k
    err = None
    own = model is None
    try:
        if own:
            model = valle_amd.VALLE(args.d_model, args.nhead, args.layers, prefix_mode=1, engine_dtype=dtype, max_batch=B)
            model.load_state_dict(sd)
            model = model.to(dev).eval()
        eng = model.engine_for(B, S, P_PROMPT)
        eng.set_option("ignore_eos", 1)  # every utterance runs to the reference's length cap (random-init weights emit EOS at arbitrary steps)
    except Exception as e:  # noqa: BLE001
        err = e
    if world > 1:
        flag = torch.tensor([0 if err is None else 1], dtype=torch.int32, device=dev)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
        if int(flag.item()) and err is None:
            err = RuntimeError("another rank failed to build this leg's model")
    if err is not None:
        raise err
    X = torch.stack([synth_inputs(rank * B + b, S)[0] for b in range(B)]).to(dev)
    Y = torch.stack([synth_inputs(rank * B + b, S)[1] for b in range(B)]).to(dev)
    s_lens, p_lens = [S] * B, [P_PROMPT] * B
    acc = dict(tokens=0, pre=0.0, ar=0.0, nar=0.0, ar_steps=0, ar_bytes=0, gl=None)

    def step():
        return decode_step(model, X, s_lens, Y, p_lens, top_k, world, world * B, dev)

    def on_step(r):
        gl, _ = r
        acc["gl"] = gl
        acc["tokens"] += sum(gl) * 8
        tm = eng.timings()
        acc["pre"] += tm["prefill_ms"]; acc["ar"] += tm["ar_ms"]; acc["nar"] += tm["nar_ms"]; acc["ar_steps"] += int(tm["ar_steps"])
        acc.setdefault("gls", []).append(list(gl))

    elapsed = timed_loop(step, steps, warmup, world, dev, on_step)
    acc["ar_bytes"] = sum(ar_bytes_of(eng, g, S) for g in acc.get("gls", []))
    tokens = acc["tokens"]
    if world > 1:
        tk = torch.tensor([tokens], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tk, op=torch.distributed.ReduceOp.SUM)
        tokens = int(tk.item())
    gl, pre, ar, nar, ar_steps, ar_bytes = acc["gl"], acc["pre"], acc["ar"], acc["nar"], acc["ar_steps"], acc["ar_bytes"]
    ps_health = {k: eng.fetch_u32("persist_" + k) for k in ("ran", "fail", "fallbacks")}
    d, L, N, G = args.d_model, args.layers, S + P_PROMPT + gl[0], gl[0]
    nar_flops = B * (7 * (2 * N * 12 * L * d * d + 4 * L * N * N * d) + 14 * G * d * 1024)  # SURVEY.md 8(d), this rank
    hbm = (ar_bytes / 1e9) / (ar / 1e3)
    tfs = nar_flops * steps / 1e12 / (nar / 1e3)
    if own:
        model._invalidate()
        del model
    mfma_peak = MFMA_PEAK_TFS if dtype != "fp32" else 157.0  # fp32 mode: v_mfma_f32_16x16x4_f32 runs at the vector rate
    res = {
        "workload": f"dim{d}-L{L}-h{args.nhead} {dtype}, batch={B} per GPU x {world} GPU(s), S={S}, P={P_PROMPT} -> G={G} frames ({G / FRAME_RATE:.2f} s), "
                    f"{'greedy (top_k=1)' if top_k == 1 else f'sampled: top_k={top_k}, temperature=1.0'}, ignore_eos, random-init weights{label}",
        "value": round(tokens / elapsed, 1), "unit": "audio-tokens/s", "n_gpus": world, "per_gpu_value": round(tokens / elapsed / world, 1),
        "steps": steps, "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 3),
        "phase_ms": {"prefill": round(pre / steps, 3), "ar": round(ar / steps, 3), "nar": round(nar / steps, 3)},
        "roofline_ar": {"bound": "hbm", "achieved": round(hbm, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm / HBM_PEAK_GBS, 4),
                        "step_us": round(ar / ar_steps * 1e3, 2), "bytes_per_step": int(ar_bytes / ar_steps), "rank": 0},
        "roofline_nar": {"bound": "mfma", "achieved": round(tfs, 1), "peak": mfma_peak, "unit": "TFLOP/s", "frac": round(tfs / mfma_peak, 4), "rank": 0},
    }
    res["persist"] = ps_health
    res.update(rates(tokens, elapsed, steps * B * world, G))
    return res


def c5_leg(args, dev, dtype="fp8", B=32, steps=2, warmup=1):
    """BASELINE.json configs[4]'s per-GPU share on one GPU: d1536-L24-h16 (dh 96), fp8 weights, 32 utterances -- engine mode
    "fp8w" (the config's weight format: e4m3 weights, bf16 activations; holds the 5 % sigma parity bar at this architecture) or
    "fp8" (+ fp8 activations on the block-scaled fp8 MFMA in the prefill / NAR passes).  Extra object, not `value`."""
    import valle_amd

    torch.manual_seed(0)
    d, L, H = 1536, 24, 16
    model = valle_amd.VALLE(d, H, L, prefix_mode=1, engine_dtype=dtype, max_batch=B).to(dev).eval()
    eng = model.engine_for(B, S_TEXT, P_PROMPT)
    eng.set_option("ignore_eos", 1)
    X = torch.stack([synth_inputs(b)[0] for b in range(B)]).to(dev)
    Y = torch.stack([synth_inputs(b)[1] for b in range(B)]).to(dev)
    s_lens, p_lens = [S_TEXT] * B, [P_PROMPT] * B

    def step():
        return decode_step(model, X, s_lens, Y, p_lens, 1, 1, B, dev)[0]

    for _ in range(warmup):
        step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    tokens, pre, ar, nar, ar_steps, ar_bytes, gls = 0, 0.0, 0.0, 0.0, 0, 0, []
    for _ in range(steps):
        gl = step()
        tokens += sum(gl) * 8
        tm = eng.timings()
        pre += tm["prefill_ms"]; ar += tm["ar_ms"]; nar += tm["nar_ms"]; ar_steps += int(tm["ar_steps"])
        gls.append(list(gl))
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    ar_bytes = sum(ar_bytes_of(eng, g, S_TEXT) for g in gls)
    N, G = S_TEXT + P_PROMPT + gl[0], gl[0]
    nar_flops = B * (7 * (2 * N * 12 * L * d * d + 4 * L * N * N * d) + 14 * G * d * 1024)
    hbm = (ar_bytes / 1e9) / (ar / 1e3)
    tfs = nar_flops * steps / 1e12 / (nar / 1e3)
    model._invalidate()
    nar_peak = 5000.0 if dtype == "fp8" else MFMA_PEAK_TFS
    what = "fp8 weights + fp8 activations on the block-scaled fp8 MFMA in prefill / NAR" if dtype == "fp8" else "fp8 (e4m3) weights, bf16 activations and MFMA"
    res = {
        "workload": f"dim{d}-L{L}-h{H} engine mode {dtype} ({what}), batch={B}, S={S_TEXT}, P={P_PROMPT} -> G={G}, greedy, ignore_eos, "
                    f"random-init weights (the reference's init distributions, torch.manual_seed(0))",
        "quoted_on": ("BASELINE configs[4] is quoted on THIS mode: the config's weight format (e4m3 weights) with bf16 activations on the bf16 MFMA; "
                      "the fp8-MFMA mode (engine mode fp8: per-row e4m3 activations, 15 % sigma bar) is experimental and measured only with --c5-fp8")
                     if dtype == "fp8w" else "experimental mode (fp8 activations, looser parity bar): not the mode configs[4] is quoted on",
        "value": round(tokens / elapsed, 1), "unit": "audio-tokens/s", "steps": steps, "warmup": warmup,
        "phase_ms": {"prefill": round(pre / steps, 3), "ar": round(ar / steps, 3), "nar": round(nar / steps, 3)},
        "roofline_ar": {"bound": "hbm", "achieved": round(hbm, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm / HBM_PEAK_GBS, 4),
                        "step_us": round(ar / ar_steps * 1e3, 2), "bytes_per_step": int(ar_bytes / ar_steps)},
        "roofline_nar": {"bound": "mfma", "achieved": round(tfs, 1), "peak": nar_peak,
                         "unit": "TFLOP/s (vs the dense fp8 MX peak)" if dtype == "fp8" else "TFLOP/s (vs the dense bf16 peak)", "frac": round(tfs / nar_peak, 4)},
    }
    res.update(rates(tokens, elapsed, steps * B, G))
    return res


def decode_step(runner, X, s_lens, Y, p_lens, top_k, world, n_total, dev, temperature=1.0, seed=0):
    """One "step" of the benchmark on this rank: whole decode of its B utterances and, for N > 1, the all_gather of the
    result codes (the path's only collective).  `runner` is the MODEL: the decode goes through `model.inference_batch()`, the
    method `VALLE.inference()` wraps (seam B2; its per-utterance EOS prints go to stderr) -- or, for the CPU tests of the
    step / timing contract, any object with the engine's prefill / generate / nar methods.  Returns (generated lengths of the
    local utterances, the gathered list of (G, 8) code matrices in global order)."""
    from valle_amd import dist as vdist

    B = X.shape[0]
    if hasattr(runner, "inference_batch"):
        import contextlib

        with contextlib.redirect_stdout(sys.stderr):
            out = runner.inference_batch(X, torch.tensor(s_lens, dtype=torch.int32), Y, p_lens, None, top_k=top_k,
                                         temperature=temperature, seed=seed)
        gl = [int(o.shape[0]) for o in out]
    else:
        if hasattr(runner, "prefill_generate"):  # the engine: one retry on VLE_EBUSY instead of an aborted timed run
            _, gl = runner.prefill_generate(X, s_lens, Y, p_lens, top_k=top_k, temperature=temperature, seed=seed, allow_empty=B > 1)
        else:
            runner.prefill(X, s_lens, Y, p_lens)
            _, gl = runner.generate(top_k=top_k, temperature=temperature, seed=seed, allow_empty=B > 1)
        codes = runner.nar(None)
        out = [codes[b, : gl[b]] for b in range(B)]
    if world > 1:
        out = vdist.gather_codes(out, n_total, 8, dev)
    return gl, out


def ar_bytes_of(eng, gl, S, sequential=False):
    """Algorithmic bytes of one decode's AR loop: per iteration W_AR*w + sum_b 2*L*d*a*(c_b + 1) over the utterances still
    generating; `sequential`: each utterance streams the weights for itself (two utterances on the batch-1 path).  753 calls into the
    library per decode: bookkeeping of the BENCHMARK, so it runs after the timed region (until round 6 it sat inside it: ~0.5 ms per step)."""
    B, tot = len(gl), 0
    if sequential:
        for b in range(B):
            for t in range(1, gl[b] + 1):
                tot += eng.ar_step_bytes(1, S + P_PROMPT + t)
    else:
        for t in range(1, max(gl) + 1):
            live = sum(1 for b in range(B) if gl[b] >= t)
            tot += eng.ar_step_bytes(live, live * (S + P_PROMPT + t))
    return tot


def rates(tokens, elapsed, n_utt_steps, frames_per_utt):
    """frames/s and the real-time factor next to a tokens/s figure: rtf = wall seconds per second of generated audio, per
    utterance stream (elapsed x streams / audio seconds); `n_utt_steps` utterance-decodes of `frames_per_utt` frames each."""
    audio_s = n_utt_steps * frames_per_utt / FRAME_RATE
    return {"frames_per_s": round(tokens / 8.0 / elapsed, 1), "audio_s_per_wall_s": round(audio_s / elapsed, 2), "rtf": round(elapsed / audio_s, 6)}


def timed_loop(step_fn, steps, warmup, world, dev, on_step=None):
    """The timing contract: W untimed warm-up steps, then exactly K steps bracketed by barrier + device sync on both
    sides; returns the MAX elapsed seconds over ranks."""
    import torch.distributed as dist

    def fence():
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)

    for _ in range(warmup):
        step_fn()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        r = step_fn()
        if on_step is not None:
            on_step(r)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start one process per GPU ourselves (what
    `python -m torch.distributed.run --nproc-per-node N` does; cf. the reference's per-device mp.spawn,
    valle/bin/trainer.py:1143-1154).  Rank 0 inherits stdout and prints the JSON line."""
    import socket
    import subprocess

    have = torch.cuda.device_count()
    if have < n:
        print(f"bench.py: --gpus {n} needs {n} visible GPUs, this box has {have} (torch.cuda.device_count()); "
              f"no CPU / oversubscribed fallback", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for pr in procs:
        rc = max(rc, abs(pr.wait()))
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=0, help="utterances per GPU of the headline `value`; default 1 at every --gpus N "
                                                         "(configs[1] per GPU: weak scaling); 64 per GPU (configs[2] / [3]) is the "
                                                         "`c3_batch64` object of every line")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "fp8w", "fp8"],
                    help="fp8w: bf16 arithmetic on fp8 e4m3 weights; fp8: fp8w + fp8 activations on the CDNA4 fp8 MFMA (configs[4])")
    ap.add_argument("--d-model", type=int, default=1024)
    ap.add_argument("--nhead", type=int, default=16)
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--top-k", type=int, default=1, help="1 = the reference's greedy; -100 = pure multinomial")
    ap.add_argument("--cpu-frames", type=int, default=128, help="frames of the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-c3", action="store_true", help="skip the extra batch-64-per-GPU (BASELINE configs[2] / [3]) object")
    ap.add_argument("--no-side", action="store_true", help="headline only: no c3_batch64 / fp32_exact / c5_share_fp8 legs (profiling runs)")
    ap.add_argument("--no-fp32", action="store_true", help="skip the extra fp32_exact object (token-exact engine mode) of the N = 1 line")
    ap.add_argument("--no-c5", action="store_true", help="skip the extra `c5_share_fp8w` object of the N = 1 line: the per-GPU share of BASELINE "
                                                         "configs[4] (d1536-L24-h16, fp8 weights, 32 utterances; ~20 s on an MI355X box "
                                                         "now that the host-side weight preparation is multi-threaded)")
    ap.add_argument("--c5-fp8", action="store_true", help="also time the configs[4] share in engine mode fp8 (looser parity bar than fp8w)")
    ap.add_argument("--c5", action="store_true", help=argparse.SUPPRESS)  # round-2 spelling: the leg is on by default now
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="engine tuning option (vle_set_option), repeatable")
    ap.add_argument("--dump-codes", default="", metavar="FILE", help="rank 0 saves the last step's gathered codes (list of (G, 8) int64 tensors, "
                                                                       "global utterance order) with torch.save: the N > 1 test compares them with N = 1 decodes")
    ap.add_argument("--profile-kernels", type=int, default=0, help="extra untimed pass: hipEvent time per AR-step kernel family over n steps")
    args = ap.parse_args()
    if args.no_side:
        args.no_c3 = args.no_fp32 = args.no_c5 = True

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    import valle_amd
    from valle_amd import dist as vdist

    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    rank, local_rank, world = vdist.env_rank_world()
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; they must agree")
    if world > torch.cuda.device_count():
        sys.exit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} visible GPUs")
    vdist.init_process_group("nccl")  # RCCL over xGMI; no-op at world size 1
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    world_seen = torch.distributed.get_world_size() if world > 1 else 1
    B = args.batch if args.batch > 0 else 1

    # random-init weights of the named architecture (no network for checkpoints), reference init distributions
    torch.manual_seed(0)
    model = valle_amd.VALLE(args.d_model, args.nhead, args.layers, prefix_mode=1, engine_dtype=args.dtype,
                            max_batch=B, use_graph=not args.no_graph)
    sd_all = {k: v.clone() for k, v in model.state_dict().items()}  # fp32 master weights: the side legs load the same ones
    sd_cpu = sd_all if (rank == 0 and args.gpus == 1 and args.cpu_frames > 0) else None
    model = model.to(dev).eval()
    eng = model.engine_for(B, S_TEXT, P_PROMPT)
    for kv in args.opt:
        name, val = kv.split("=")
        eng.set_option(name, int(val))
    # random-init weights emit EOS at arbitrary steps for some seeds (utterance 23 at step 0); every configuration is timed to the
    # reference's length cap (16 S + 1 frames; the greedy batch-1 run reaches it anyway)
    eng.set_option("ignore_eos", 1)

    X = torch.zeros(B, S_TEXT, dtype=torch.int64)
    Y = torch.zeros(B, P_PROMPT, 8, dtype=torch.int64)
    for b in range(B):
        X[b], Y[b] = synth_inputs(rank * B + b)
    X, Y = X.to(dev), Y.to(dev)
    s_lens, p_lens = [S_TEXT] * B, [P_PROMPT] * B

    acc = dict(tokens=0, pre=0.0, ar=0.0, nar=0.0, ar_steps=0, ar_bytes=0, gl=None, n_out=0)

    def step():
        return decode_step(model, X, s_lens, Y, p_lens, args.top_k, world, world * B, dev)

    def on_step(r):
        gl, out = r
        acc["gl"], acc["n_out"] = gl, len(out)
        acc["out"] = out
        acc["tokens"] += sum(gl) * 8
        seq = getattr(model, "sequential_timings", None)  # two utterances decoded one after the other on the batch-1 path (model.py)
        tm = seq if seq is not None else eng.timings()
        acc["sequential"] = seq is not None
        acc["pre"] += tm["prefill_ms"]; acc["ar"] += tm["ar_ms"]; acc["nar"] += tm["nar_ms"]; acc["ar_steps"] += int(tm["ar_steps"])
        acc.setdefault("gls", []).append(list(gl))

    elapsed = timed_loop(step, args.steps, args.warmup, world, dev, on_step)
    acc["ar_bytes"] = sum(ar_bytes_of(eng, g, S_TEXT, bool(acc.get("sequential"))) for g in acc.get("gls", []))
    tokens, gl = acc["tokens"], acc["gl"]
    pre_ms, ar_ms, nar_ms, ar_steps, ar_bytes = acc["pre"], acc["ar"], acc["nar"], acc["ar_steps"], acc["ar_bytes"]
    assert acc["n_out"] == world * B, "the gather did not return every rank's utterances"
    if args.dump_codes and rank == 0:
        torch.save([o.cpu() for o in acc["out"]], args.dump_codes)
    if world > 1:
        tk = torch.tensor([tokens], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tk, op=torch.distributed.ReduceOp.SUM)
        tokens = int(tk.item())

    kernel_prof = None
    if args.profile_kernels > 0 and rank == 0:
        eng.set_option("profile_kernels", args.profile_kernels)
        eng.prefill(X, s_lens, Y, p_lens)
        eng.generate(top_k=args.top_k, max_new=args.profile_kernels)
        kernel_prof = eng.kernel_times()
        eng.set_option("profile_kernels", 0)

    out = None
    if rank == 0:
        step_ms = ar_ms / max(ar_steps, 1)
        achieved = (ar_bytes / 1e9) / (ar_ms / 1e3) if ar_ms > 0 else 0.0
        traffic, traffic_src = measured_traffic(args, B)
        fused = args.d_model // args.nhead in (64, 128) and B == 1 and "qkv_attn=0" not in args.opt
        # the persistent launch's health over the timed decodes: `ran` = the last AR loop ran pstep_kernel (not merely "would"), `fail` =
        # waves that gave up in it, `fallbacks` = calls since engine creation that ended with VLE_EBUSY and were repeated on the chain
        ps_health = {"active": eng.fetch_u32("persist_active"), "ran": eng.fetch_u32("persist_ran"), "fail": eng.fetch_u32("persist_fail"),
                     "fallbacks": eng.fetch_u32("persist_fallbacks"), "backoff": eng.fetch_u32("persist_backoff")}
        persist = ps_health["ran"] == 1 if B == 1 else False
        own_sample = persist and eng.fetch_u32("persist_sample_active") == 1
        # kernel launches of the dominant kernel over the timed decodes: with the sampling step inside the persistent launch one launch
        # runs several AR iterations (the engine's last call is representative: every timed decode has the same length)
        launches = eng.fetch_u32("ar_launches") * args.steps if own_sample else ar_steps
        n1_ref = None
        try:
            with open(N1_REF_FILE) as f:
                n1_ref = json.load(f)
        except OSError:
            pass
        out = {
            "metric": "audio-tokens/sec (AR+NAR decode, 3 s prompt -> 10 s target)",
            "value": round(tokens / elapsed, 1),
            "unit": "audio-tokens/s",
            "n_gpus": args.gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": f"dim{args.d_model}-L{args.layers}-h{args.nhead} {args.dtype}, batch={B} per GPU, S={S_TEXT} text tokens, "
                            f"P={P_PROMPT} prompt frames (3 s) -> G={gl[0]} frames ({gl[0] / 75:.2f} s), "
                            f"{'greedy (top_k=1)' if args.top_k == 1 else f'top_k={args.top_k}'}, random-init weights",
                "parallelism": f"batch-sharded x{args.gpus} (independent utterances, gather of codes only)",
                "batch_per_gpu": B,
                "sequential": bool(acc.get("sequential")),  # two utterances decoded one after the other on the persistent batch-1 launch (model.py)
                "world_size": world_seen,
                "backend": "nccl (RCCL)" if world > 1 else "none (single process)",
                "hip_graph": not args.no_graph,
                "persist": ps_health,
            },
            "per_gpu_value": round(tokens / elapsed / args.gpus, 1),
            **rates(tokens, elapsed, args.steps * B * world, gl[0]),
            # weak scaling in numbers: per-GPU work is fixed at every N; efficiency = value(N) / (N x value(1)) against the committed
            # N = 1 line (profiles/bench_n1_reference.json) -- 1.0 by definition at N = 1; the c3_batch64 efficiency is filled in below
            "scale_ref": {
                "n_gpus": args.gpus, "utterances_per_gpu": B, "c3_utterances_per_gpu": 64,
                "value_n1": (n1_ref or {}).get("value"), "c3_value_n1": (n1_ref or {}).get("c3_value"),
                "n1_source": (n1_ref or {}).get("source"),
                "efficiency": 1.0 if args.gpus == 1 else (round(tokens / elapsed / (args.gpus * n1_ref["value"]), 4) if n1_ref and n1_ref.get("value") else None),
                "c3_efficiency": None,
            },
            "phase_ms": {"prefill": round(pre_ms / args.steps, 3), "ar": round(ar_ms / args.steps, 3), "nar": round(nar_ms / args.steps, 3)},
            "roofline": {
                "kernel": ("pstep_kernel = persistent AR launch (256 workgroups; per iteration L layers + final norm + predict layer + sampling, stop "
                           "rule and next-token embedding, in-launch granule hand-offs; several AR iterations per launch, hipGraph replay); per "
                           "iteration the weights + KV are streamed once") if own_sample else
                          ("AR decode step = ONE persistent launch (pstep_kernel: 256 workgroups, L layers + final norm + predict layer, in-launch "
                           "granule hand-offs) + the sampling launch, hipGraph replay; weights + KV streamed once") if persist else
                          ("AR decode step (hipGraph replay: 4 launches/layer x L -- fused LN1+QKV+attention, out-proj, FFN1, FFN2 -- + logits + sample; "
                           "weights + KV streamed once)") if fused else
                          "AR decode step (hipGraph replay: launches/layer x L + logits + sample; weights + KV streamed once)",
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                # HBM bytes per AR step from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs, FETCH_SIZE
                # doubled per the gfx950 correction), read from profiles/ar_step_traffic.json and only reported while the step's
                # kernels are the ones it was measured on (kernel-set hash); null + the reason otherwise
                "traffic": None if traffic is None else int(traffic * ar_steps / max(launches, 1)),  # per launch, like `achieved`
                "traffic_per_step": traffic,
                "traffic_source": traffic_src,
                # per launch of the kernel (what rocprofv3's average duration is compared with) and per AR iteration
                "launch_us": round(ar_ms / max(launches, 1) * 1e3, 2),
                "bytes_per_launch": int(ar_bytes / max(launches, 1)),
                "launches": launches,
                "iterations": ar_steps,
                "iterations_per_launch": round(ar_steps / max(launches, 1), 2),
                "step_us": round(step_ms * 1e3, 2),
                "bytes_per_step": int(ar_bytes / max(ar_steps, 1)),
            },
        }
        if kernel_prof is not None:
            out["roofline"]["kernel_us"] = kernel_prof
        # the side legs must never cost the headline line: report their failure instead of raising
        if sd_cpu is not None:
            try:
                out["cpu_baseline"] = cpu_baseline(sd_cpu, args.d_model, args.nhead, args.layers, args.cpu_frames)
            except Exception as err:  # noqa: BLE001
                out["cpu_baseline"] = {"error": repr(err)[:200]}
        else:  # N > 1 (or --cpu-frames 0): the committed N = 1 measurement, with its provenance
            try:
                with open(CPU_CACHE_FILE) as f:
                    out["cpu_baseline"] = dict(json.load(f), cached=True, source="profiles/cpu_baseline_n1.json (the `cpu_baseline` object of a "
                                               "default N = 1 run of this bench.py on an MI355X box; timed on rank 0 at N = 1 only)")
            except OSError:
                out["cpu_baseline"] = None
        if sd_cpu is not None:
            try:
                model._invalidate()
                out["eager_gpu_baseline"] = eager_gpu_baseline(sd_cpu, args.d_model, args.nhead, args.layers, args.cpu_frames, dev)
            except Exception as err:  # noqa: BLE001
                out["eager_gpu_baseline"] = {"error": repr(err)[:200]}
    # ---- side legs: every rank takes part (same timing contract as the headline) ---------------------------------------
    plain = B == 1 and not args.opt and args.profile_kernels == 0 and args.dtype == "bf16"

    def leg(name, fn, need_all_ranks):
        """world = 1: a failing leg is reported, never raised (it must not cost the headline line).  N > 1: legs with collectives let
        the exception propagate -- the launcher then ends every rank instead of leaving the others in a barrier."""
        if need_all_ranks and world > 1:
            res = fn()
        else:
            try:
                res = fn()
            except Exception as err:  # noqa: BLE001
                res = {"error": repr(err)[:200]}
        if out is not None:
            out[name] = res
        return res

    if plain and world == 1 and not args.no_side:
        # the reference's own default sampling (valle/models/valle.py:967-968) and a realistic text length, on the headline model
        leg("sampled", lambda: side_leg(sd_all, args, dev, rank, world, 1, args.dtype, steps=3, warmup=1, top_k=-100, model=model), False)
        leg("s200", lambda: side_leg(sd_all, args, dev, rank, world, 1, args.dtype, steps=2, warmup=1, S=200, model=model,
                                     label="; SURVEY.md 8(d) 'realistic text' row"), False)
    model._invalidate()
    if plain and world == 1 and not args.no_side:
        # 4 utterances share ONE persistent launch (csrc/persist_nb.hip, round 6): the small-batch serving point between the headline and c3
        leg("batch4", lambda: side_leg(sd_all, args, dev, rank, world, 4, args.dtype, steps=3, warmup=1,
                                       label="; 2 .. 6 utterances run the batched persistent AR launch (weights streamed once per step for all of them)"), False)
    if plain and not args.no_c3:
        c3 = leg("c3_batch64", lambda: side_leg(sd_all, args, dev, rank, world, 64, args.dtype), True)
        if out is not None and isinstance(c3, dict) and "value" in c3:
            ref = out["scale_ref"].get("c3_value_n1")
            out["scale_ref"]["c3_efficiency"] = 1.0 if args.gpus == 1 else (round(c3["value"] / (args.gpus * ref), 4) if ref else None)
    if plain and not args.no_fp32 and world == 1:
        leg("fp32_exact", lambda: side_leg(sd_all, args, dev, rank, world, 1, "fp32", steps=3, warmup=1,
                                           label="; engine mode fp32 = greedy token ids bit-identical to the reference (tests/test_parity_sizes_gpu.py)"), False)
    if plain and world == 1 and not args.no_c5 and (args.d_model, args.layers, args.nhead) == (1024, 12, 16):
        del model
        # BASELINE configs[4] is quoted on fp8w -- the config's WEIGHT format, which holds the 5 % sigma parity bar at this architecture.
        # Engine mode fp8 (per-row e4m3 activations on the block-scaled MFMA) is held to a looser bar (15 % sigma max / 3 % mean,
        # tests/test_parity_sizes_gpu.py) and its MX per-32-column activation scales were not built: it is measured only on request.
        leg("c5_share_fp8w", lambda: c5_leg(args, dev, "fp8w"), False)
        if args.c5_fp8:
            leg("c5_share_fp8", lambda: c5_leg(args, dev, "fp8"), False)
    if out is not None:
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
