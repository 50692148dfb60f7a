"""TEST INFRASTRUCTURE ONLY -- generate tests/golden/*.npz by running the UNMODIFIED
reference (``/root/reference``, imported via oracle/ref_import.py) on CPU fp32.

Run in the build container (the reference does not exist on the GPU box):

    python oracle/make_golden.py [--only NAME] [--skip-large]

Weights come from ``oracle.valle_oracle.make_state_dict(cfg, seed)`` (loaded into the
reference model with ``load_state_dict(strict=True)`` -- which also proves the state-dict
contract of SURVEY.md 8a) and inputs from ``make_inputs(S, P, seed)``; both are
regenerated deterministically by the tests, so a fixture only stores the reference's
OUTPUTS: final codes, per-step AR logits (captured by rebinding
``valle.models.valle.topk_sampling`` from outside) and NAR-stage
logits (forward hooks on ``nar_predict_layers``).
"""
from __future__ import annotations

import argparse
import hashlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import valle_oracle as vo  # noqa: E402
from oracle.ref_import import AttributeDict, import_reference  # noqa: E402

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

# name -> dict(cfg kwargs, S, P, wseed, iseed, mode, top_k, enroll, ar_stride, nar_rows)
CASES = {
    # reference smoke-test shapes (valle/tests/valle_test.py:91-135): d=64, h=16 => dh=4
    "tiny_dh4_pm1": dict(cfg=dict(d_model=64, nhead=16, num_layers=2, prefix_mode=1), S=6, P=10, ar_stride=8),
    "tiny_bos_pm0": dict(cfg=dict(d_model=64, nhead=4, num_layers=2, prefix_mode=0, prepend_bos=True), S=5, P=9, ar_stride=8),
    "tiny_pm2_enroll": dict(cfg=dict(d_model=128, nhead=4, num_layers=3, prefix_mode=2), S=9, P=14, enroll=4, ar_stride=8),
    "tiny_pm4_q6": dict(cfg=dict(d_model=128, nhead=2, num_layers=2, prefix_mode=4, num_quantizers=6), S=8, P=11, enroll=3, ar_stride=8),
    "tiny_q1": dict(cfg=dict(d_model=64, nhead=2, num_layers=2, prefix_mode=1, num_quantizers=1), S=4, P=8, ar_stride=8),
    "tiny_noshare": dict(cfg=dict(d_model=64, nhead=1, num_layers=1, prefix_mode=1, share_embedding=False), S=4, P=6, ar_stride=8),
    "tiny_continual": dict(cfg=dict(d_model=64, nhead=4, num_layers=2, prefix_mode=1), S=7, P=40, mode="continual"),
    "tiny_continual_pm0": dict(cfg=dict(d_model=64, nhead=4, num_layers=2, prefix_mode=0), S=7, P=31, mode="continual"),
    # constructor options outside the fused engine's shape (post-norm layers, prenets, a different NAR width: the reference's own
    # smoke test runs them, valle/tests/valle_test.py:106-133) -- decoded by the block-module path of valle_amd/model.py
    "opt_postnorm_prenet_pm1": dict(cfg=dict(d_model=64, nhead=4, num_layers=2, norm_first=False, add_prenet=True, prefix_mode=1), S=6, P=10, ar_stride=8),
    "opt_postnorm_prenet_half_bos_pm0": dict(cfg=dict(d_model=64, nhead=4, num_layers=2, norm_first=False, add_prenet=True, prefix_mode=0,
                                                      nar_scale_factor=0.5, prepend_bos=True), S=5, P=9, ar_stride=8),
    "opt_prenorm_prenet_pm1": dict(cfg=dict(d_model=64, nhead=2, num_layers=2, add_prenet=True, prefix_mode=1), S=6, P=10, ar_stride=8),
    "opt_postnorm_pm2_enroll": dict(cfg=dict(d_model=128, nhead=4, num_layers=2, norm_first=False, prefix_mode=2), S=9, P=14, enroll=4, ar_stride=8),
    "opt_scale2_pm1": dict(cfg=dict(d_model=64, nhead=2, num_layers=1, nar_scale_factor=2.0, prefix_mode=1), S=5, P=8, ar_stride=8),
    "opt_postnorm_prenet_continual": dict(cfg=dict(d_model=64, nhead=4, num_layers=2, norm_first=False, add_prenet=True, prefix_mode=1), S=7, P=40,
                                          mode="continual"),
    # VALL-F (valle.py:50-710): the text as cross-attention memory of nn.TransformerDecoder layers
    "vallf_prenorm_pm1": dict(cfg=dict(d_model=64, nhead=4, num_layers=2, prefix_mode=1, model="vallf"), S=6, P=10, ar_stride=8),
    "vallf_postnorm_prenet_bos_pm0": dict(cfg=dict(d_model=64, nhead=2, num_layers=2, norm_first=False, add_prenet=True, prefix_mode=0,
                                                   prepend_bos=True, model="vallf"), S=5, P=9, ar_stride=8),
    "vallf_pm2_enroll_half": dict(cfg=dict(d_model=128, nhead=4, num_layers=2, prefix_mode=2, nar_scale_factor=0.5, model="vallf"), S=9, P=14,
                                  enroll=4, ar_stride=8),
    "small_dh64": dict(cfg=dict(d_model=128, nhead=2, num_layers=2, prefix_mode=1), S=12, P=30, ar_stride=8),
    "small_dh96": dict(cfg=dict(d_model=192, nhead=2, num_layers=2, prefix_mode=1), S=10, P=20, ar_stride=8),
    # BASELINE.json configs[0]: dim256-L6-h4, S=47, P=225 -> G=753 (the CPU-runnable plumbing case)
    "c1_d256_L6": dict(cfg=dict(d_model=256, nhead=4, num_layers=6, prefix_mode=1), S=47, P=225, ar_stride=16, large=True),
    # BASELINE.json configs[1] architecture, shortened so the no-KV-cache reference finishes in minutes
    "c2_d1024_L12_short": dict(cfg=dict(d_model=1024, nhead=16, num_layers=12, prefix_mode=1), S=16, P=75, ar_stride=8, large=True),
    # BASELINE.json configs[1] at FULL size (the benchmark's own shape: S=47, P=225 -> G=753); ~10-20 min of the
    # no-KV-cache reference on 8 threads.  ar_all: every step's logits are kept (fp16) so the bf16 engine can be
    # teacher-forced against the reference over the whole run.
    "c2_d1024_L12_full": dict(cfg=dict(d_model=1024, nhead=16, num_layers=12, prefix_mode=1), S=47, P=225, ar_stride=16, nar_rows=24,
                              ar_all=True, large=True, huge=True),
}


# VALL-F builds its decoders with torch's nn.TransformerDecoder (valle.py:141); the container of torch >= 2 rejects the
# reference's tuple inputs and passes keywords its layers do not take.  For the VALL-F fixtures ONLY the
# container's forward is replaced, from outside, by the loop of torch 1.13.1 (the reference's pin, README.md:31): each layer
# on (output, memory), then the norm.  Every layer, norm, embedding and the inference / forward code are the reference's own.
def dec_forward_113(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None, memory_key_padding_mask=None, **_):
    output = tgt
    for mod in self.layers:
        output = mod(output, memory, tgt_mask=tgt_mask, memory_mask=memory_mask, tgt_key_padding_mask=tgt_key_padding_mask,
                     memory_key_padding_mask=memory_key_padding_mask)
    if self.norm is not None:
        output = self.norm(output)
    return output


def build_reference(vm, cfg: vo.OracleConfig, sd):
    p = AttributeDict(
        model_name="vall-f" if cfg.model == "vallf" else "valle", decoder_dim=cfg.d_model, nhead=cfg.nhead, num_decoder_layers=cfg.num_layers,
        norm_first=cfg.norm_first, add_prenet=cfg.add_prenet, prefix_mode=cfg.prefix_mode,
        share_embedding=cfg.share_embedding, scale_factor=cfg.nar_scale_factor,
        prepend_bos=cfg.prepend_bos, num_quantizers=cfg.num_quantizers,
    )
    model = vm.get_model(p).eval()
    assert list(model.state_dict().keys()) == list(sd.keys()), "state-dict key contract differs from reference"
    model.load_state_dict(sd, strict=True)
    return model


def run_case(vm, name: str, spec: dict):
    import valle.models.valle as ref_valle  # the reference module whose sampler we wrap

    cfg = vo.OracleConfig(**spec["cfg"])
    wseed, iseed = spec.get("wseed", 0), spec.get("iseed", 1234)
    S, P = spec["S"], spec["P"]
    sd = vo.make_state_dict(cfg, wseed)
    x, x_lens, y = vo.make_inputs(S, P, iseed, Q=cfg.num_quantizers)
    enroll = torch.tensor([spec["enroll"]], dtype=torch.int32) if "enroll" in spec else None
    model = build_reference(vm, cfg, sd)

    ar_logits = []
    orig = ref_valle.topk_sampling

    def spy(logits, top_k=10, top_p=1.0, temperature=1.0):
        ar_logits.append(logits.detach().clone()[0])
        return orig(logits, top_k=top_k, top_p=top_p, temperature=temperature)

    nar_logits = {}
    hooks = []
    if cfg.num_quantizers > 1:
        for i, layer in enumerate(model.nar_predict_layers):
            hooks.append(layer.register_forward_hook(lambda m, a, out, i=i: nar_logits.__setitem__(i, out.detach().clone()[0])))

    orig_dec_forward = torch.nn.TransformerDecoder.forward
    if cfg.model == "vallf":
        torch.nn.TransformerDecoder.forward = dec_forward_113  # see its comment: torch 1.13's container loop, nothing else
    ref_valle.topk_sampling = spy
    t0 = time.time()
    try:
        with torch.no_grad():
            if spec.get("mode") == "continual":
                codes = model.continual(x, x_lens, y)
            else:
                codes = model.inference(x, x_lens, y, enroll_x_lens=enroll, top_k=spec.get("top_k", 1), temperature=1.0)
    finally:
        ref_valle.topk_sampling = orig
        torch.nn.TransformerDecoder.forward = orig_dec_forward
        for h in hooks:
            h.remove()
    wall = time.time() - t0

    out = dict(
        codes=codes[0].numpy().astype(np.int16),
        S=np.int32(S), P=np.int32(P), wseed=np.int32(wseed), iseed=np.int32(iseed),
        enroll=np.int32(spec.get("enroll", -1)),
        mode=np.bytes_(spec.get("mode", "inference")),
        top_k=np.int32(spec.get("top_k", 1)),
        ref_wall_s=np.float32(wall),
        torch_version=np.bytes_(torch.__version__),
    )
    for k, v in spec["cfg"].items():
        out[f"cfg_{k}"] = np.asarray(v)
    if ar_logits:
        al = torch.stack(ar_logits)  # (G+1, 1025): one per loop iteration incl. the stopping one
        top2 = torch.topk(al, 2, dim=-1)[0]
        out["ar_margin"] = (top2[:, 0] - top2[:, 1]).numpy().astype(np.float32)
        out["ar_logit_std"] = np.float32(al.std().item())
        stride = spec.get("ar_stride", 1)
        out["ar_stride"] = np.int32(stride)
        out["ar_logits"] = al[::stride].numpy().astype(np.float32)
        if spec.get("ar_all"):
            out["ar_logits_all_f16"] = al.numpy().astype(np.float16)  # |err| <= 2^-11 |logit|: far below the bf16 bar
    if nar_logits:
        rows = min(spec.get("nar_rows", 3), nar_logits[0].shape[0])
        idx = np.linspace(0, nar_logits[0].shape[0] - 1, rows).astype(np.int64)
        out["nar_rows"] = idx
        out["nar_logits"] = np.stack([nar_logits[i][idx].numpy() for i in sorted(nar_logits)]).astype(np.float32)
        m = []
        for i in sorted(nar_logits):
            t2 = torch.topk(nar_logits[i], 2, dim=-1)[0]
            m.append((t2[:, 0] - t2[:, 1]).numpy())
        out["nar_margin"] = np.stack(m).astype(np.float32)
        out["nar_logit_std"] = np.asarray([nar_logits[i].std().item() for i in sorted(nar_logits)], dtype=np.float32)
    sha = hashlib.sha256(codes.numpy().astype(np.int64).tobytes()).hexdigest()[:16]
    out["codes_sha256_16"] = np.bytes_(sha)
    path = os.path.join(GOLDEN_DIR, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"[golden] {name}: codes {tuple(codes.shape)} sha {sha} ref wall {wall:.1f}s -> {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--skip-large", action="store_true")
    ap.add_argument("--threads", type=int, default=8)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    vm = import_reference()
    for name, spec in CASES.items():
        if args.only and name != args.only:
            continue
        if args.skip_large and spec.get("large"):
            continue
        if spec.get("huge") and args.only != name:
            continue  # only on request: python oracle/make_golden.py --only c2_d1024_L12_full
        run_case(vm, name, spec)


if __name__ == "__main__":
    main()
