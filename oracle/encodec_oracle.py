"""TEST INFRASTRUCTURE ONLY -- CPU oracle of the EnCodec 24 kHz DECODER (codes -> waveform), the step right after the
hot path: ``audio_tokenizer.decode([(codes.transpose(2, 1), None)])`` (valle/bin/infer.py:261-263,
valle/data/tokenizer.py:241-242 -> ``EncodecModel.decode``).

Parity status: **UNPINNED**.  The algorithm lives in the third-party ``encodec`` package (un-pinned in the reference:
setup.py:109 ``encodec``; valle/data/tokenizer.py:224 ``EncodecModel.encodec_model_24khz()``, 6 kbps = 8 codebooks).  It is
not installed here, is not vendored by the reference, and its pretrained weights cannot be fetched (no network); the
reference holds no golden vector for it.  This file restates the PUBLISHED architecture of that model (encodec/model.py,
encodec/modules/seanet.py ``SEANetDecoder``, encodec/modules/conv.py, encodec/modules/lstm.py, encodec/quantization/
{vq,core_vq}.py, v0.1.1) with torch's own conv / LSTM ops, on deterministic synthetic weights in the package's state-dict
naming; the HIP implementation (valle_amd/csrc/codec.hip) is checked against it.  Whoever has the real checkpoint can pin
both by loading ``EncodecModel.encodec_model_24khz().state_dict()`` into ``decode`` below and into ``valle_amd.codec``.

Architecture (encodec_model_24khz: SEANetDecoder(channels=1, dimension=128, n_filters=32, n_residual_layers=1,
ratios=[8, 5, 4, 2], activation=ELU(alpha=1), norm="weight_norm", kernel_size=7, last_kernel_size=7,
residual_kernel_size=3, dilation_base=2, causal=True, pad_mode="reflect", true_skip=False, compress=2, lstm=2,
trim_right_ratio=1.0)):
  RVQ decode      sum_q codebook_q[codes[:, q]]                       (T, 128)          quantization/core_vq.py decode
  model.0         SConv1d(128 -> 512, k 7, causal reflect pad)
  model.1         SLSTM(512, 2 layers) with skip: y = LSTM(x) + x      modules/lstm.py
  model.3i+2..4   for ratio r in 8, 5, 4, 2 (channels c: 512 -> 256 -> 128 -> 64 -> 32):
                    ELU; SConvTranspose1d(c -> c/2, k 2r, stride r, causal: trim the right k - r samples);
                    SEANetResnetBlock(c/2): shortcut SConv1d(k 1)(x) + [ELU, SConv1d(c/2 -> c/4, k 3), ELU, SConv1d(c/4 -> c/2, k 1)](x)
  model.14, 15    ELU; SConv1d(32 -> 1, k 7)                           -> (1, 320 T) samples at 24 kHz
"""
from __future__ import annotations

import hashlib
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

RATIOS = [8, 5, 4, 2]
DIM, NF, BINS = 128, 32, 1024
HOP = 320  # prod(RATIOS): samples per frame (75 Hz at 24 kHz)


def state_dict_spec(n_q: int = 8) -> "OrderedDict[str, tuple]":
    """Decoder-side keys of EncodecModel.encodec_model_24khz().state_dict() (weight_norm stored as weight_g / weight_v)."""
    s: "OrderedDict[str, tuple]" = OrderedDict()

    def conv(prefix, cout, cin, k):
        s[f"{prefix}.conv.conv.weight_g"] = (cout, 1, 1)
        s[f"{prefix}.conv.conv.weight_v"] = (cout, cin, k)
        s[f"{prefix}.conv.conv.bias"] = (cout,)

    for q in range(n_q):
        s[f"quantizer.vq.layers.{q}._codebook.embed"] = (BINS, DIM)
    c = NF * 2 ** len(RATIOS)  # 512
    conv("decoder.model.0", c, DIM, 7)
    for layer in range(2):
        s[f"decoder.model.1.lstm.weight_ih_l{layer}"] = (4 * c, c)
        s[f"decoder.model.1.lstm.weight_hh_l{layer}"] = (4 * c, c)
        s[f"decoder.model.1.lstm.bias_ih_l{layer}"] = (4 * c,)
        s[f"decoder.model.1.lstm.bias_hh_l{layer}"] = (4 * c,)
    idx = 2
    for r in RATIOS:
        s[f"decoder.model.{idx + 1}.convtr.convtr.weight_g"] = (c, 1, 1)       # ConvTranspose1d weight is (in, out, k): norm over dim 0
        s[f"decoder.model.{idx + 1}.convtr.convtr.weight_v"] = (c, c // 2, 2 * r)
        s[f"decoder.model.{idx + 1}.convtr.convtr.bias"] = (c // 2,)
        conv(f"decoder.model.{idx + 2}.block.1", c // 4, c // 2, 3)
        conv(f"decoder.model.{idx + 2}.block.3", c // 2, c // 4, 1)
        conv(f"decoder.model.{idx + 2}.shortcut", c // 2, c // 2, 1)
        c //= 2
        idx += 3
    conv(f"decoder.model.{idx + 1}", 1, NF, 7)  # model.15
    return s


def make_state_dict(seed: int = 0, n_q: int = 8) -> "OrderedDict[str, torch.Tensor]":
    """Deterministic synthetic weights (kaiming-like scales so activations stay O(1) through the stack)."""
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for key, shape in state_dict_spec(n_q).items():
        g = torch.Generator().manual_seed(int.from_bytes(hashlib.sha256(f"{seed}:{key}".encode()).digest()[:7], "little"))
        if key.endswith("embed"):
            t = torch.randn(shape, generator=g) * 0.5
        elif key.endswith("weight_g"):
            t = 0.6 + 0.8 * torch.rand(shape, generator=g)
        elif key.endswith("weight_v"):
            t = torch.randn(shape, generator=g)
        elif "lstm.weight" in key:
            t = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(shape[1])
        else:  # biases
            t = (torch.rand(shape, generator=g) * 2 - 1) * 0.1
        sd[key] = t.float().contiguous()
    # weight_norm makes ||w_slice|| = g: shrink g with the fan-in so that O(1) inputs stay O(1) through the stack
    for key in list(sd):
        if key.endswith("weight_g"):
            fan = sd[key[:-1] + "v"][0].numel()
            sd[key] = sd[key] * (1.4 if "convtr" in key else 1.0) * 8.0 / math.sqrt(max(fan, 8))
    return sd


def fold_weight_norm(sd, prefix: str) -> torch.Tensor:
    """torch.nn.utils.weight_norm (dim 0): w = g * v / ||v|| with the norm over every dim but 0.  Also accepts a plain
    ``weight`` or the parametrization naming of newer torch."""
    if f"{prefix}.weight" in sd:
        return sd[f"{prefix}.weight"]
    if f"{prefix}.parametrizations.weight.original0" in sd:
        g, v = sd[f"{prefix}.parametrizations.weight.original0"], sd[f"{prefix}.parametrizations.weight.original1"]
    else:
        g, v = sd[f"{prefix}.weight_g"], sd[f"{prefix}.weight_v"]
    return v * (g / v.flatten(1).norm(dim=1).view(-1, 1, 1))


def _pad_left_reflect(x: torch.Tensor, pad: int) -> torch.Tensor:
    """encodec.modules.conv.pad1d(x, (pad, 0), mode="reflect"): inputs not longer than the pad are first zero-extended on
    the right to pad + 1 samples, reflected, then cut back."""
    if pad == 0:
        return x
    T = x.shape[-1]
    extra = 0
    if T <= pad:
        extra = pad - T + 1
        x = F.pad(x, (0, extra))
    y = F.pad(x, (pad, 0), mode="reflect")
    return y[..., : y.shape[-1] - extra]


def sconv1d(sd, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """SConv1d, causal, stride 1, dilation 1 (modules/conv.py): reflect-pad k - 1 on the left, then Conv1d.  x (C, T)."""
    w = fold_weight_norm(sd, f"{prefix}.conv.conv")
    return F.conv1d(_pad_left_reflect(x[None], w.shape[-1] - 1), w, sd[f"{prefix}.conv.conv.bias"])[0]


def sconvtr1d(sd, prefix: str, x: torch.Tensor, stride: int) -> torch.Tensor:
    """SConvTranspose1d, causal, trim_right_ratio 1: ConvTranspose1d then drop the last k - stride samples."""
    w = fold_weight_norm(sd, f"{prefix}.convtr.convtr")
    y = F.conv_transpose1d(x[None], w, sd[f"{prefix}.convtr.convtr.bias"], stride=stride)[0]
    return y[..., : y.shape[-1] - (w.shape[-1] - stride)]


def slstm(sd, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """SLSTM (modules/lstm.py): 2-layer nn.LSTM over time, plus the skip connection.  x (C, T)."""
    C = x.shape[0]
    lstm = torch.nn.LSTM(C, C, 2)
    with torch.no_grad():
        for name, p in lstm.named_parameters():
            p.copy_(sd[f"{prefix}.lstm.{name}"])
    y, _ = lstm(x.t()[:, None, :])  # (T, 1, C)
    return y[:, 0, :].t() + x


def rvq_decode(sd, codes: torch.Tensor) -> torch.Tensor:
    """ResidualVectorQuantization.decode (quantization/core_vq.py): sum over codebooks of embed[codes].  codes (T, Q) -> (128, T)."""
    out = torch.zeros(codes.shape[0], DIM)
    for q in range(codes.shape[1]):
        out = out + F.embedding(codes[:, q], sd[f"quantizer.vq.layers.{q}._codebook.embed"])
    return out.t().contiguous()


@torch.no_grad()
def decode(sd, codes: torch.Tensor, trace=None) -> torch.Tensor:
    """EncodecModel.decode for one utterance without the (disabled at 24 kHz) scale: codes int64 (T, Q) -> wav fp32 (320 T,)."""
    assert codes.dim() == 2 and codes.dtype == torch.int64
    x = rvq_decode(sd, codes)
    x = sconv1d(sd, "decoder.model.0", x)
    x = slstm(sd, "decoder.model.1", x)
    if trace is not None:
        trace["lstm"] = x.clone()
    idx = 2
    for r in RATIOS:
        x = sconvtr1d(sd, f"decoder.model.{idx + 1}", F.elu(x), r)
        p = f"decoder.model.{idx + 2}"
        h = sconv1d(sd, f"{p}.block.1", F.elu(x))
        h = sconv1d(sd, f"{p}.block.3", F.elu(h))
        x = sconv1d(sd, f"{p}.shortcut", x) + h
        if trace is not None:
            trace[f"stage{r}"] = x.clone()
        idx += 3
    x = sconv1d(sd, f"decoder.model.{idx + 1}", F.elu(x))
    return x[0]
