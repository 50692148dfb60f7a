"""Block API of the hot path ("Transformer module surface", SURVEY.md 8b seam B3) on the HIP operators.

Same class names, constructor arguments, parameter names (state-dict keys), call signatures and
tuple-in/tuple-out conventions as the reference's

* ``TokenEmbedding`` / ``SinePositionalEmbedding``       valle/modules/embedding.py:21-97
* ``LayerNorm`` / ``AdaptiveLayerNorm``                  valle/modules/transformer.py:17-108
* ``MultiheadAttention``                                 valle/modules/activation.py:18-431
* ``TransformerEncoderLayer`` / ``TransformerEncoder``   valle/modules/transformer.py:178-406

so that ``model.ar_decoder((x, None), mask=m)`` / ``model.nar_decoder((x, stage_emb))``
(valle/models/valle.py:1035-1038, 1125-1127) read the same against this package.  Every ``forward``
runs ``vle_op_*`` kernels of ``libvalle_engine.so`` on a ROCm device; there is no CPU or PyTorch
arithmetic path -- a CPU tensor raises.  ``VALLE.inference()`` itself does not go through these
modules (it drives the fused, KV-cached, graph-captured engine); they are the per-block surface of
the same kernels, checked against the reference's modules in tests/test_modules_gpu.py.

Only what the decode path uses is implemented; everything else raises ``NotImplementedError``:
pre-norm and post-norm layers, ReLU, self-attention with ``attn_mask`` either ``None`` or the
prefix-LM / causal pattern of valle.py:1019-1033, no key-padding, eval mode (dropout = identity; BatchNorm1d of
the prenets on its running statistics).
"""
from __future__ import annotations

import copy
import math
import numbers
from typing import Any, Callable, Optional, Tuple, Union

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops

Tensor = torch.Tensor


def _need_device(t: Tensor, who: str):
    if not t.is_cuda:
        raise RuntimeError(f"{who}: the HIP operators need tensors on a ROCm device (there is no CPU path)")


class _HipModule(nn.Module):
    """Parameters stay fp32 (the reference's state dict); ``compute_dtype`` selects the element type the
    GEMM / attention kernels run in ("fp32": token-exact mode, "bf16": bf16 operands, fp32 accumulate,
    "fp8w": bf16 kernels on the fp8-representable weights W' of the engine's FP8W mode)."""

    compute_dtype: str = "fp32"

    def _tdtype(self) -> torch.dtype:
        return torch.float32 if self.compute_dtype == "fp32" else torch.bfloat16

    def _w(self, p: Tensor) -> Tensor:
        """The parameter in the compute dtype (bf16 copies are cached until the parameter changes); in "fp8w"
        mode the bf16 copy of W' = e4m3fn(w / 2^e) * 2^e, the values the engine's FP8W mode computes with."""
        if self.compute_dtype == "fp32":
            return p.detach()
        cache = self.__dict__.setdefault("_wcache", {})
        key = id(p)
        hit = cache.get(key)
        if hit is not None and hit[0] == (p._version, p.data_ptr()):
            return hit[1]
        if self.compute_dtype == "fp8w" and p.dim() == 2:
            w = ops.quantize_fp8w(p)[2].to(p.device, torch.bfloat16)  # host quantiser of the engine, once per weight
        else:
            w = p.detach().to(torch.bfloat16)
        cache[key] = ((p._version, p.data_ptr()), w)
        return w


def set_compute_dtype(module: nn.Module, dtype: str) -> nn.Module:
    """Select "fp32" or "bf16" kernels for every block module under ``module``."""
    assert dtype in ("fp32", "bf16", "fp8w"), dtype
    for m in module.modules():
        if isinstance(m, _HipModule):
            m.compute_dtype = dtype
            m.__dict__.pop("_wcache", None)
    return module


# ---- valle/modules/embedding.py --------------------------------------------------------------------
class TokenEmbedding(_HipModule):
    """valle/modules/embedding.py:21-47."""

    def __init__(self, dim_model: int, vocab_size: int, dropout: float = 0.0):
        super().__init__()
        self.vocab_size = vocab_size
        self.dim_model = dim_model
        self.dropout = nn.Dropout(p=dropout)
        self.word_embeddings = nn.Embedding(self.vocab_size, self.dim_model)

    @property
    def weight(self) -> Tensor:
        return self.word_embeddings.weight

    def embedding(self, index: int) -> Tensor:
        return self.word_embeddings.weight[index: index + 1]

    def forward(self, x: Tensor) -> Tensor:
        _need_device(self.word_embeddings.weight, "TokenEmbedding")
        if self.training and self.dropout.p > 0:
            raise NotImplementedError("dropout (training) is outside the decode path")
        return ops.token_embedding(x.to(self.word_embeddings.weight.device, torch.int64), self.word_embeddings.weight.detach())

    def add_to(self, inout: Tensor, x: Tensor) -> Tensor:
        """``inout += self(x)`` in one kernel (no reference counterpart as a method: it is ``y_emb[...] += embedding_layer(codes)``,
        valle.py:1104-1113, 1134); ``inout`` a contiguous fp32 view of whole rows."""
        _need_device(self.word_embeddings.weight, "TokenEmbedding")
        return ops.token_embedding_add(inout, x.to(self.word_embeddings.weight.device, torch.int64), self.word_embeddings.weight.detach())


def sine_pe_table(length: int, dim_model: int) -> Tensor:
    """The reference's table, built by the same fp32 torch ops (embedding.py:75-91), (length, d) on the host."""
    pe = torch.zeros(length, dim_model)
    position = torch.arange(0, length, dtype=torch.float32).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, dim_model, 2, dtype=torch.float32) * -(math.log(10000.0) / dim_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


class SinePositionalEmbedding(_HipModule):
    """valle/modules/embedding.py:50-97 (``reverse`` is never set by the reference; not carried)."""

    def __init__(self, dim_model: int, dropout: float = 0.0, scale: bool = False, alpha: bool = False):
        super().__init__()
        self.dim_model = dim_model
        self.x_scale = math.sqrt(dim_model) if scale else 1.0
        self.alpha = nn.Parameter(torch.ones(1), requires_grad=alpha)
        self.dropout = nn.Dropout(p=dropout)
        self.pe: Optional[Tensor] = None  # (1, T, d) like the reference's attribute (not a buffer: not in the state dict)

    def extend_pe(self, x: Tensor):
        T = x.size(1)
        if self.pe is not None and self.pe.size(1) >= T and self.pe.device == x.device:
            return
        self.pe = sine_pe_table(max(T, 4000), self.dim_model).unsqueeze(0).to(x.device)

    def forward(self, x: Tensor) -> Tensor:
        _need_device(x, "SinePositionalEmbedding")
        if self.training and self.dropout.p > 0:
            raise NotImplementedError("dropout (training) is outside the decode path")
        self.extend_pe(x)
        xin = x.unsqueeze(-1) if x.ndim == 2 else x
        if xin.shape[-1] != self.dim_model:
            xin = xin.expand(*xin.shape[:-1], self.dim_model)
        return ops.sine_positional(xin.to(torch.float32), self.pe[0], self.alpha.detach(), self.x_scale)


# ---- valle/models/valle.py: prenets ----------------------------------------------------------------------
class Transpose(nn.Identity):
    """valle/models/valle.py:38-42, (N, T, D) -> (N, D, T).  A place holder inside the prenet containers below (their
    forward works time-major and never calls it); kept so that the nn.Sequential indices -- the state-dict keys -- match."""

    def forward(self, input: Tensor) -> Tensor:
        return input.transpose(1, 2)


class TextPrenet(nn.Sequential):
    """``{ar,nar}_text_prenet`` (valle/models/valle.py:100-116, 183-206): 3 x [Conv1d(d, d, 5, padding="same") -> BatchNorm1d ->
    ReLU -> Dropout(0.5)] over time, then Linear(d, d).  Same children at the same indices as the reference (parameters and
    BatchNorm buffers keep their keys: 1, 2, 5, 6, 9, 10, 14).  Eval mode only.  Each conv block is ONE fp32 GEMM: the five
    taps of the zero-padded, time-major signal are the K dimension (window rows of 5 d contiguous values), BatchNorm's running
    statistics and affine are folded into the conv's weight and bias, ReLU is the GEMM epilogue."""

    def __init__(self, d: int):
        blocks = []
        for _ in range(3):
            blocks += [nn.Conv1d(d, d, kernel_size=5, padding="same"), nn.BatchNorm1d(d), nn.ReLU(), nn.Dropout(0.5)]
        super().__init__(Transpose(), *blocks, Transpose(), nn.Linear(d, d))
        self.d = d

    def _folded(self, conv: nn.Conv1d, bn: nn.BatchNorm1d):
        cache = self.__dict__.setdefault("_fold", {})
        key = id(conv)
        ver = tuple((t._version, t.data_ptr()) for t in (conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var))
        hit = cache.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1], hit[2]
        # parameter preparation (once per weight version), in fp64: y = (conv(x) + b - mean) * g / sqrt(var + eps) + beta
        sc = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
        w = (conv.weight.detach().double() * sc[:, None, None]).permute(0, 2, 1).reshape(self.d, 5 * self.d)  # [out][tap * d + in]
        b = (conv.bias.detach().double() - bn.running_mean.detach().double()) * sc + bn.bias.detach().double()
        w, b = w.float().contiguous(), b.float().contiguous()
        cache[key] = (ver, w, b)
        return w, b

    def forward(self, x: Tensor) -> Tensor:
        _need_device(x, "TextPrenet")
        if self.training:
            raise NotImplementedError("the prenets run in eval mode only (BatchNorm running statistics, no dropout)")
        N, T, d = x.shape
        h = x.to(torch.float32)
        for conv_i, bn_i in ((1, 2), (5, 6), (9, 10)):
            w, b = self._folded(self[conv_i], self[bn_i])
            hp = torch.nn.functional.pad(h, (0, 0, 2, 2))                       # zero "same" padding along time (data movement)
            win = hp.unfold(1, 5, 1).permute(0, 1, 3, 2).reshape(N * T, 5 * d)  # row t = taps t .. t+4, each d values
            h = ops.linear(win.contiguous(), w, b, epilogue=ops.EPI_RELU).view(N, T, d)
        lin = self[14]
        return ops.linear(h.reshape(N * T, d), lin.weight.detach(), lin.bias.detach(), epilogue=ops.EPI_F32).view(N, T, d)


class AudioPrenet(nn.Sequential):
    """``{ar,nar}_audio_prenet`` (valle/models/valle.py:118-126, 207-215): Linear(d, 256) ReLU Dropout Linear(256, 256) ReLU
    Dropout Linear(256, d) per frame (keys 0, 3, 6): three fp32 GEMMs."""

    def __init__(self, d: int):
        super().__init__(nn.Linear(d, 256), nn.ReLU(), nn.Dropout(0.25), nn.Linear(256, 256), nn.ReLU(), nn.Dropout(0.25), nn.Linear(256, d))

    def forward(self, y: Tensor) -> Tensor:
        _need_device(y, "AudioPrenet")
        if self.training:
            raise NotImplementedError("the prenets run in eval mode only")
        shape = y.shape
        h = y.to(torch.float32).reshape(-1, shape[-1]).contiguous()
        h = ops.linear(h, self[0].weight.detach(), self[0].bias.detach(), epilogue=ops.EPI_RELU)
        h = ops.linear(h, self[3].weight.detach(), self[3].bias.detach(), epilogue=ops.EPI_RELU)
        return ops.linear(h, self[6].weight.detach(), self[6].bias.detach(), epilogue=ops.EPI_F32).view(shape)


# ---- valle/modules/transformer.py: norms -------------------------------------------------------------
class LayerNorm(_HipModule):
    """valle/modules/transformer.py:17-80: F.layer_norm over the last dim; tuple in => tuple out."""

    def __init__(self, normalized_shape, eps: float = 1e-5, elementwise_affine: bool = True, device=None, dtype=None):
        super().__init__()
        if isinstance(normalized_shape, numbers.Integral):
            normalized_shape = (normalized_shape,)
        self.normalized_shape = tuple(normalized_shape)
        if len(self.normalized_shape) != 1:
            raise NotImplementedError("LayerNorm over more than the last dim is not on the decode path")
        if eps != 1e-5:
            raise NotImplementedError("the kernels fix eps = 1e-5 (transformer.py:27)")
        if not elementwise_affine:
            raise NotImplementedError("elementwise_affine=False is not on the decode path")
        self.eps = eps
        self.elementwise_affine = elementwise_affine
        self.weight = nn.Parameter(torch.ones(self.normalized_shape, device=device))
        self.bias = nn.Parameter(torch.zeros(self.normalized_shape, device=device))

    def _norm(self, x: Tensor, out_dtype: torch.dtype, gamma: Optional[Tensor] = None, beta: Optional[Tensor] = None) -> Tensor:
        _need_device(x, "LayerNorm")
        g = self.weight.detach() if gamma is None else gamma
        b = self.bias.detach() if beta is None else beta
        return ops.layernorm(x.to(torch.float32), g, b, out_dtype=out_dtype)

    def forward(self, input: Union[Tensor, Tuple[Tensor, Any]], embedding: Any = None):
        if isinstance(input, tuple):
            input, embedding = input
            return (self._norm(input, torch.float32), embedding)
        assert embedding is None
        return self._norm(input, torch.float32)

    def extra_repr(self) -> str:
        return "{normalized_shape}, eps={eps}, elementwise_affine={elementwise_affine}".format(**self.__dict__)


class AdaptiveLayerNorm(_HipModule):
    """valle/modules/transformer.py:83-108: ``[w, b] = project_layer(embedding); w * norm(input) + b``,
    evaluated as ONE LayerNorm launch with the folded affine (w * g, w * be + b) -- the same fold
    ``vle_finalize_weights`` applies per NAR stage."""

    def __init__(self, d_model: int, norm: LayerNorm) -> None:
        super().__init__()
        self.project_layer = nn.Linear(d_model, 2 * d_model)
        self.norm = norm
        self.d_model = d_model
        self.eps = self.norm.eps

    def _norm(self, x: Tensor, embedding: Tensor, out_dtype: torch.dtype) -> Tensor:
        _need_device(x, "AdaptiveLayerNorm")
        if embedding is None or embedding.numel() != self.d_model:
            raise NotImplementedError("AdaptiveLayerNorm needs ONE stage embedding of d_model values (valle.py:1125)")
        emb = embedding.detach().reshape(1, self.d_model).to(torch.float32)
        # the projection of one row: fp32 weight-streaming GEMV (input-independent, 2d x d)
        wb = ops.linear_skinny(emb, self.project_layer.weight.detach(), self.project_layer.bias.detach())
        gamma, beta = ops.adaln_fold(wb.reshape(-1), self.norm.weight.detach(), self.norm.bias.detach())
        return self.norm._norm(x, out_dtype, gamma, beta)

    def forward(self, input: Union[Tensor, Tuple[Tensor, Tensor]], embedding: Tensor = None):
        if isinstance(input, tuple):
            input, embedding = input
            return (self._norm(input, embedding, torch.float32), embedding)
        return self._norm(input, embedding, torch.float32)


# ---- valle/modules/activation.py ---------------------------------------------------------------------
def classify_attn_mask(attn_mask: Optional[Tensor], T: int) -> Tuple[int, bool]:
    """Map an ``attn_mask`` to the two mask classes the attention kernels implement:
    returns (text_len, causal) such that row i sees keys j < max(text_len, i + 1 if causal else T).

    * ``None``                                    -> (0, False)   no mask (NAR, valle.py:1125)
    * bool (T, T), True = blocked; float, -inf = blocked: must be the prefix-LM pattern of
      valle.py:1019-1033 (first ``S`` rows see exactly the first S keys, later rows are causal; S = 0 or 1
      is the plain causal mask)                   -> (S, True)
    Anything else raises NotImplementedError (costs one device sync -- this is the block API, the engine
    never materialises a mask)."""
    if attn_mask is None:
        return 0, False
    if attn_mask.dim() != 2 or tuple(attn_mask.shape) != (T, T):
        raise NotImplementedError(f"attn_mask of shape {tuple(attn_mask.shape)}: only (T, T) masks are implemented")
    if attn_mask.dtype == torch.bool:
        blocked = attn_mask
    elif attn_mask.is_floating_point():
        blocked = torch.isneginf(attn_mask)
        if not bool(((attn_mask == 0) | blocked).all()):
            raise NotImplementedError("additive float masks other than 0 / -inf are not implemented")
    else:
        raise NotImplementedError(f"attn_mask dtype {attn_mask.dtype}")
    S = int((~blocked[0]).sum().item())  # row 0 sees keys j < max(S, 1)
    i = torch.arange(T, device=blocked.device)
    want_allowed = i[None, :] < torch.maximum(i[:, None] + 1, torch.tensor(S, device=blocked.device))
    if not torch.equal(~blocked, want_allowed):
        raise NotImplementedError("attn_mask is neither None nor the prefix-LM / causal pattern of valle.py:1019-1033")
    return S, True


class MultiheadAttention(_HipModule):
    """valle/modules/activation.py:18-431, self-attention form used by TransformerEncoderLayer._sa_block
    (transformer.py:315-329): packed in-proj [Q;K;V], heads = contiguous dh-slices, out_proj."""

    def __init__(self, embed_dim, num_heads, dropout=0.0, bias=True, add_bias_kv=False, add_zero_attn=False, kdim=None,
                 vdim=None, batch_first=False, linear1_cls=nn.Linear, linear2_cls=nn.Linear, device=None, dtype=None) -> None:
        super().__init__()
        if add_bias_kv or add_zero_attn or (kdim not in (None, embed_dim)) or (vdim not in (None, embed_dim)) or not bias:
            raise NotImplementedError("only the packed self-attention configuration of VALL-E is implemented")
        if linear1_cls is not nn.Linear or linear2_cls is not nn.Linear:
            raise NotImplementedError("scaled linears (scaling.py) are outside the decode path")
        self.embed_dim = embed_dim
        self.kdim = self.vdim = embed_dim
        self._qkv_same_embed_dim = True
        self.num_heads = num_heads
        self.dropout = dropout
        self.batch_first = batch_first
        self.head_dim = embed_dim // num_heads
        assert self.head_dim * num_heads == self.embed_dim, "embed_dim must be divisible by num_heads"
        self.in_proj_weight = nn.Parameter(torch.empty((3 * embed_dim, embed_dim), device=device))
        self.in_proj_bias = nn.Parameter(torch.empty(3 * embed_dim, device=device))
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=True, device=device)
        self.bias_k = self.bias_v = None
        self.add_zero_attn = False
        self._reset_parameters()

    def _reset_parameters(self):  # activation.py:175-190
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.in_proj_bias, 0.0)
        nn.init.constant_(self.out_proj.bias, 0.0)

    def _attend(self, xn: Tensor, B: int, T: int, attn_mask: Optional[Tensor], key_padding_mask: Optional[Tensor] = None) -> Tensor:
        """xn (B*T, d) in the compute dtype -> attention output (B*T, d), before out_proj.

        ``key_padding_mask`` (B, T) bool, True = padded position (``src_key_padding_mask`` of the teacher-forced forward,
        valle.py:846-856, :908-926; trailing padding per segment -- VALL-E's [text | text pad | audio | audio pad] -- anything else raises): padded KEYS are invisible to
        everyone.  Valid rows are packed per sequence and attend under ``attn_mask`` restricted to the valid keys (the ragged
        packed-row kernel: positions keep their order, the prefix-LM text length counts valid text rows).  Padded QUERY rows are
        real rows of the reference's batch (its AR loss sums over them): each sees exactly the valid keys of its sequence -- every
        valid key precedes a padded audio row, so that is what the prefix-LM / causal mask leaves it, and for padded text rows
        (never read by anyone: masked as keys, not part of the logits) it is merely finite -- computed by the un-masked
        cross-attention kernel against the sequence's valid K / V."""
        text_len, causal = classify_attn_mask(attn_mask, T)
        pad = None
        if key_padding_mask is not None and bool(key_padding_mask.any()):
            if key_padding_mask.dtype != torch.bool:
                raise NotImplementedError("key_padding_mask must be a bool mask (True = padded); float masks are not implemented")
            pad = key_padding_mask.to(xn.device)
            assert tuple(pad.shape) == (B, T), (tuple(pad.shape), B, T)
            if bool(pad.all(dim=1).any()):
                raise ValueError("key_padding_mask leaves a sequence without any key")
            # padded QUERY rows are given every valid key of their sequence: that is what the reference computes only when padding
            # is TRAILING inside each segment ([text | pad | audio | pad], the collater's shape); anything else would diverge silently
            # (without an attention mask -- the NAR decoder -- every query sees every valid key whatever the pattern: nothing to check)
            segments = () if (not causal and text_len == 0) else (((0, text_len), (text_len, T)) if text_len > 0 else ((0, T),))
            for lo, hi in segments:
                seg = pad[:, lo:hi]
                if seg.shape[1] > 1 and bool((seg[:, :-1] & ~seg[:, 1:]).any()):
                    raise NotImplementedError("key_padding_mask: only trailing padding per segment ([text | pad | audio | pad]) is implemented")
        qkv = ops.linear(xn, self._w(self.in_proj_weight), self.in_proj_bias.detach())
        if pad is None:
            seq_off = torch.arange(0, (B + 1) * T, T, dtype=torch.int32, device=xn.device)
            tl = torch.full((B,), text_len, dtype=torch.int32, device=xn.device)
            return ops.attention(qkv, seq_off, tl, self.num_heads, causal)
        d = qkv.shape[1] // 3
        valid = ~pad
        flat_valid = valid.reshape(-1).nonzero(as_tuple=False).squeeze(1)
        n_valid = valid.sum(dim=1)
        seq_off = torch.zeros(B + 1, dtype=torch.int32, device=xn.device)
        seq_off[1:] = n_valid.cumsum(0).to(torch.int32)
        tl = valid[:, :text_len].sum(dim=1).to(torch.int32) if text_len > 0 else torch.zeros(B, dtype=torch.int32, device=xn.device)
        att = torch.empty(B * T, d, dtype=qkv.dtype, device=xn.device)
        att[flat_valid] = ops.attention(qkv[flat_valid].contiguous(), seq_off, tl, self.num_heads, causal)
        offs = seq_off.tolist()
        for b in range(B):  # padded query rows: all valid keys of their own sequence (scoring batches: a few rows each)
            prow = pad[b].nonzero(as_tuple=False).squeeze(1) + b * T
            if prow.numel() == 0:
                continue
            vrow = flat_valid[offs[b]: offs[b + 1]]
            att[prow] = ops.cross_attention(qkv[prow, :d].contiguous(), qkv[vrow, d:].contiguous(), self.num_heads)
        return att

    def _cross_forward(self, query: Tensor, key: Tensor, value: Tensor, key_padding_mask, need_weights, attn_mask):
        """MultiheadAttention.forward(x, mem, mem): VALL-F's ``multihead_attn`` (valle/modules/transformer.py:582-597): queries
        projected by the first d rows of the packed in-proj, keys / values of the memory by the other 2 d (activation.py:128-130),
        un-masked softmax(Q K^T / sqrt(dh)) V per head (``vle_op_cross_attention``), out_proj."""
        if value is not key:
            raise NotImplementedError("cross-attention with different key and value tensors is not on the decode path")
        if attn_mask is not None:
            raise NotImplementedError("memory_mask: VALL-F passes None (valle.py:629)")
        if key_padding_mask is not None and bool(key_padding_mask.any()):
            raise NotImplementedError("memory_key_padding_mask: the decode path runs unpadded sequences")
        if need_weights:
            raise NotImplementedError("need_weights=True is not produced (transformer.py:594 calls with need_weights=False)")
        if self.training and self.dropout > 0:
            raise NotImplementedError("dropout (training) is outside the decode path")
        x, mem = query, key
        unbatched = x.dim() == 2
        if unbatched:
            x, mem = x.unsqueeze(0 if self.batch_first else 1), mem.unsqueeze(0 if self.batch_first else 1)
        if not self.batch_first:
            x, mem = x.transpose(0, 1), mem.transpose(0, 1)
        B, T, d = x.shape
        S = mem.shape[1]
        tdt = self._tdtype()
        w, b = self._w(self.in_proj_weight), self.in_proj_bias.detach()
        q = ops.linear(x.reshape(B * T, d).to(tdt).contiguous(), w[:d], b[:d])
        kv = ops.linear(mem.reshape(B * S, d).to(tdt).contiguous(), w[d:], b[d:])
        att = torch.cat([ops.cross_attention(q[i * T:(i + 1) * T], kv[i * S:(i + 1) * S], self.num_heads) for i in range(B)], dim=0)
        out = ops.linear(att, self._w(self.out_proj.weight), self.out_proj.bias.detach(), epilogue=ops.EPI_F32).view(B, T, d)
        if not self.batch_first:
            out = out.transpose(0, 1)
        if unbatched:
            out = out.squeeze(0 if self.batch_first else 1)
        return out, None

    def forward(self, query: Tensor, key: Tensor, value: Tensor, key_padding_mask: Optional[Tensor] = None,
                need_weights: bool = True, attn_mask: Optional[Tensor] = None, average_attn_weights: bool = True):
        _need_device(query, "MultiheadAttention")
        if key is not query:
            return self._cross_forward(query, key, value, key_padding_mask, need_weights, attn_mask)
        if value is not query:
            raise NotImplementedError("self-attention with a separate value tensor is not on the decode path")
        if need_weights:
            raise NotImplementedError("need_weights=True (attention maps) is not produced by the flash kernels; "
                                      "the decode path calls with need_weights=False (transformer.py:327)")
        if self.training and self.dropout > 0:
            raise NotImplementedError("dropout (training) is outside the decode path")
        x = query
        unbatched = x.dim() == 2
        if unbatched:
            x = x.unsqueeze(0 if self.batch_first else 1)
        if not self.batch_first:
            x = x.transpose(0, 1)
        B, T, d = x.shape
        xt = x.reshape(B * T, d).to(self._tdtype()).contiguous()
        att = self._attend(xt, B, T, attn_mask, key_padding_mask)
        out = ops.linear(att, self._w(self.out_proj.weight), self.out_proj.bias.detach(), epilogue=ops.EPI_F32).view(B, T, d)
        if not self.batch_first:
            out = out.transpose(0, 1)
        if unbatched:
            out = out.squeeze(0 if self.batch_first else 1)
        return out, None



# ---- valle/modules/transformer.py: encoder -----------------------------------------------------------
class TransformerEncoderLayer(_HipModule):
    """valle/modules/transformer.py:178-334.  Pre-norm branch (:296-302):
    ``x += SA(norm1(x)); x += W2 relu(W1 norm2(x) + b1) + b2`` -- LayerNorm, QKV GEMM, attention, out-proj GEMM
    (+residual), LayerNorm, FFN1 GEMM (+ReLU), FFN2 GEMM (+residual).  Post-norm branch (:303-308):
    ``x = norm1(x + SA(x)); x = norm2(x + FFN(x))`` -- the same kernels, the norms after the residual GEMMs."""

    def __init__(self, d_model: int, nhead: int, dim_feedforward: int = 2048, dropout: float = 0.1,
                 activation: Union[str, Callable[[Tensor], Tensor]] = F.relu, batch_first: bool = False, norm_first: bool = False,
                 device=None, dtype=None, linear1_self_attention_cls=nn.Linear, linear2_self_attention_cls=nn.Linear,
                 linear1_feedforward_cls=nn.Linear, linear2_feedforward_cls=nn.Linear, layer_norm_cls=LayerNorm,
                 layer_norm_eps: float = 1e-5, adaptive_layer_norm=False) -> None:
        super().__init__()
        if not (activation is F.relu or activation == "relu" or isinstance(activation, nn.ReLU)):
            raise NotImplementedError("only ReLU is fused in the FFN1 epilogue (VALL-E passes no activation: transformer.py:187)")
        if linear1_feedforward_cls is not nn.Linear or linear2_feedforward_cls is not nn.Linear or layer_norm_cls is not LayerNorm:
            raise NotImplementedError("scaled linears / BasicNorm (scaling.py) are outside the decode path")
        self.self_attn = MultiheadAttention(d_model, nhead, dropout=dropout, batch_first=batch_first,
                                            linear1_cls=linear1_self_attention_cls, linear2_cls=linear2_self_attention_cls, device=device)
        self.linear1 = nn.Linear(d_model, dim_feedforward, device=device)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model, device=device)
        self.norm_first = norm_first
        self.batch_first = batch_first
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.activation = F.relu
        norm1 = LayerNorm(d_model, eps=layer_norm_eps, device=device)
        norm2 = LayerNorm(d_model, eps=layer_norm_eps, device=device)
        if adaptive_layer_norm:
            self.norm1 = AdaptiveLayerNorm(d_model, norm1)
            self.norm2 = AdaptiveLayerNorm(d_model, norm2)
        else:
            self.norm1 = norm1
            self.norm2 = norm2

    def _n(self, norm, x2: Tensor, stage_embedding) -> Tensor:
        if isinstance(norm, AdaptiveLayerNorm):
            return norm._norm(x2, stage_embedding, self._tdtype())
        assert stage_embedding is None  # LayerNorm.forward's own assertion (transformer.py:69)
        return norm._norm(x2, self._tdtype())

    def forward(self, src, src_mask: Optional[Tensor] = None, src_key_padding_mask: Optional[Tensor] = None):
        x, stage_embedding = src, None
        is_src_tuple = isinstance(src, tuple)
        if is_src_tuple:
            x, stage_embedding = src
        _need_device(x, "TransformerEncoderLayer")
        if src_key_padding_mask is not None:
            if src_key_padding_mask.dtype != torch.bool and not torch.is_floating_point(src_key_padding_mask):
                raise AssertionError("only bool and floating types of key_padding_mask are supported")
            if src_key_padding_mask.dtype != torch.bool:
                raise NotImplementedError("floating key_padding_mask: the teacher-forced forward passes bool masks (make_pad_mask)")
        if self.training and self.dropout.p > 0:
            raise NotImplementedError("dropout (training) is outside the decode path")
        xb = x if self.batch_first else x.transpose(0, 1)
        B, T, d = xb.shape
        res = xb.to(torch.float32).reshape(B * T, d).clone()  # fp32 residual stream; the GEMM epilogues add into it
        if not self.norm_first:
            return self._post_norm(res, B, T, d, src_mask, stage_embedding, is_src_tuple, src_key_padding_mask)
        xn = self._n(self.norm1, res, stage_embedding)
        att = self.self_attn._attend(xn, B, T, src_mask, src_key_padding_mask)
        sa = self.self_attn
        ops.linear(att, sa._w(sa.out_proj.weight), sa.out_proj.bias.detach(), epilogue=ops.EPI_RESID, resid=res)
        xn = self._n(self.norm2, res, stage_embedding)
        h = ops.linear(xn, self._w(self.linear1.weight), self.linear1.bias.detach(), epilogue=ops.EPI_RELU)
        ops.linear(h, self._w(self.linear2.weight), self.linear2.bias.detach(), epilogue=ops.EPI_RESID, resid=res)
        out = res.view(B, T, d)
        if not self.batch_first:
            out = out.transpose(0, 1)
        return (out, stage_embedding) if is_src_tuple else out


    def _n32(self, norm, x2: Tensor, stage_embedding) -> Tensor:
        if isinstance(norm, AdaptiveLayerNorm):
            return norm._norm(x2, stage_embedding, torch.float32)
        assert stage_embedding is None
        return norm._norm(x2, torch.float32)

    def _post_norm(self, res: Tensor, B: int, T: int, d: int, src_mask, stage_embedding, is_src_tuple: bool, key_padding_mask=None):
        """transformer.py:303-308: x = norm1(x + SA(x)); x = norm2(x + W2 relu(W1 x + b1) + b2)."""
        sa = self.self_attn
        tdt = self._tdtype()
        att = sa._attend(res if tdt == torch.float32 else res.to(tdt), B, T, src_mask, key_padding_mask)
        ops.linear(att, sa._w(sa.out_proj.weight), sa.out_proj.bias.detach(), epilogue=ops.EPI_RESID, resid=res)  # res = x + SA(x)
        x1 = self._n32(self.norm1, res, stage_embedding)                                                       # the new residual stream
        x1c = x1 if tdt == torch.float32 else self._n(self.norm1, res, stage_embedding)                         # ... in the GEMM's element type
        h = ops.linear(x1c, self._w(self.linear1.weight), self.linear1.bias.detach(), epilogue=ops.EPI_RELU)
        ops.linear(h, self._w(self.linear2.weight), self.linear2.bias.detach(), epilogue=ops.EPI_RESID, resid=x1)  # x1 += FFN(x1)
        out = self._n32(self.norm2, x1, stage_embedding).view(B, T, d)
        if not self.batch_first:
            out = out.transpose(0, 1)
        return (out, stage_embedding) if is_src_tuple else out


def _get_clones(module, N):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(N)])


class TransformerEncoder(_HipModule):
    """valle/modules/transformer.py:337-406: N deep-copied layers + optional final norm."""

    def __init__(self, encoder_layer, num_layers, norm=None):
        super().__init__()
        self.layers = _get_clones(encoder_layer, num_layers)
        self.num_layers = num_layers
        self.norm = norm

    def forward(self, src, mask: Optional[Tensor] = None, src_key_padding_mask: Optional[Tensor] = None,
                return_layer_states: bool = False):
        if return_layer_states:
            layer_states = []
            output = src
            for mod in self.layers:
                output = mod(output, src_mask=mask, src_key_padding_mask=src_key_padding_mask)
                layer_states.append(output[0])  # transformer.py:389 (first element of the tuple / first batch row)
            if self.norm is not None:
                output = self.norm(output)
            return layer_states, output
        output = src
        for mod in self.layers:
            output = mod(output, src_mask=mask, src_key_padding_mask=src_key_padding_mask)
        if self.norm is not None:
            output = self.norm(output)
        return output


# ---- valle/modules/transformer.py: decoder (VALL-F) ---------------------------------------------------
class TransformerDecoderLayer(_HipModule):
    """valle/modules/transformer.py:409-616: self-attention over the (audio) target, cross-attention over the (text) memory, FFN;
    pre-norm (:544-556) or post-norm (:557-571); plain or adaptive LayerNorms norm1..3.  The same kernels as the encoder layer
    plus ``vle_op_cross_attention``; the residual stream stays fp32 and the residual GEMM epilogues add into it."""

    def __init__(self, d_model: int, nhead: int, dim_feedforward: int = 2048, dropout: float = 0.1,
                 activation: Union[str, Callable[[Tensor], Tensor]] = F.relu, linear1_self_attention_cls=nn.Linear,
                 linear2_self_attention_cls=nn.Linear, linear1_feedforward_cls=nn.Linear, linear2_feedforward_cls=nn.Linear,
                 batch_first: bool = False, norm_first: bool = False, device=None, dtype=None, layer_norm_cls=LayerNorm,
                 layer_norm_eps: float = 1e-5, adaptive_layer_norm=False) -> None:
        super().__init__()
        if not (activation is F.relu or activation == "relu" or isinstance(activation, nn.ReLU)):
            raise NotImplementedError("only ReLU is fused in the FFN1 epilogue")
        if linear1_feedforward_cls is not nn.Linear or linear2_feedforward_cls is not nn.Linear or layer_norm_cls is not LayerNorm:
            raise NotImplementedError("scaled linears / BasicNorm (scaling.py) are outside the decode path")
        mk = lambda: MultiheadAttention(d_model, nhead, dropout=dropout, batch_first=batch_first, linear1_cls=linear1_self_attention_cls,  # noqa: E731
                                        linear2_cls=linear2_self_attention_cls, device=device)
        self.self_attn = mk()
        self.multihead_attn = mk()
        self.linear1 = nn.Linear(d_model, dim_feedforward, device=device)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model, device=device)
        self.norm_first = norm_first
        self.batch_first = batch_first
        self.dropout1, self.dropout2, self.dropout3 = nn.Dropout(dropout), nn.Dropout(dropout), nn.Dropout(dropout)
        self.activation = F.relu
        norms = [LayerNorm(d_model, eps=layer_norm_eps, device=device) for _ in range(3)]
        if adaptive_layer_norm:
            norms = [AdaptiveLayerNorm(d_model, n) for n in norms]
        self.norm1, self.norm2, self.norm3 = norms

    def _n(self, norm, x2: Tensor, stage_embedding, out_dtype: torch.dtype) -> Tensor:
        if isinstance(norm, AdaptiveLayerNorm):
            return norm._norm(x2, stage_embedding, out_dtype)
        assert stage_embedding is None
        return norm._norm(x2, out_dtype)

    def _cross(self, xq: Tensor, mem_kv: Tensor, B: int, T: int, S: int) -> Tensor:
        """xq (B*T, d) in the compute dtype -> cross-attention output before out_proj; mem_kv (B*S, 2d) projected memory."""
        ca = self.multihead_attn
        d = xq.shape[1]
        q = ops.linear(xq, ca._w(ca.in_proj_weight)[:d], ca.in_proj_bias.detach()[:d])
        return torch.cat([ops.cross_attention(q[i * T:(i + 1) * T], mem_kv[i * S:(i + 1) * S], ca.num_heads) for i in range(B)], dim=0)

    def forward(self, tgt, memory: Tensor, tgt_mask: Optional[Tensor] = None, memory_mask: Optional[Tensor] = None,
                tgt_key_padding_mask: Optional[Tensor] = None, memory_key_padding_mask: Optional[Tensor] = None):
        x, stage_embedding = tgt, None
        tgt_is_tuple = isinstance(tgt, tuple)
        if tgt_is_tuple:
            x, stage_embedding = tgt
        _need_device(x, "TransformerDecoderLayer")
        if memory_mask is not None:
            raise NotImplementedError("memory_mask: VALL-F passes None (valle.py:629)")
        for pm in (tgt_key_padding_mask, memory_key_padding_mask):
            if pm is not None and bool(pm.to(torch.bool).any()):
                raise NotImplementedError("key_padding_mask: the decode path runs unpadded sequences")
        if self.training and self.dropout.p > 0:
            raise NotImplementedError("dropout (training) is outside the decode path")
        xb = x if self.batch_first else x.transpose(0, 1)
        mb = memory if self.batch_first else memory.transpose(0, 1)
        B, T, d = xb.shape
        S = mb.shape[1]
        tdt = self._tdtype()
        sa, ca = self.self_attn, self.multihead_attn
        lin = lambda a, w, b, **k: ops.linear(a, self._w(w), b.detach(), **k)  # noqa: E731
        # the memory's keys / values: rows d .. 3d of the packed in-proj (activation.py:128-130)
        mem_kv = ops.linear(mb.reshape(B * S, d).to(tdt).contiguous(), ca._w(ca.in_proj_weight)[d:], ca.in_proj_bias.detach()[d:])
        res = xb.to(torch.float32).reshape(B * T, d).clone()
        if self.norm_first:  # :544-556
            att = sa._attend(self._n(self.norm1, res, stage_embedding, tdt), B, T, tgt_mask)
            lin(att, sa.out_proj.weight, sa.out_proj.bias, epilogue=ops.EPI_RESID, resid=res)
            catt = self._cross(self._n(self.norm2, res, stage_embedding, tdt), mem_kv, B, T, S)
            lin(catt, ca.out_proj.weight, ca.out_proj.bias, epilogue=ops.EPI_RESID, resid=res)
            h = lin(self._n(self.norm3, res, stage_embedding, tdt), self.linear1.weight, self.linear1.bias, epilogue=ops.EPI_RELU)
            lin(h, self.linear2.weight, self.linear2.bias, epilogue=ops.EPI_RESID, resid=res)
            out = res
        else:  # :557-571
            cast = (lambda t: t) if tdt == torch.float32 else (lambda t: t.to(tdt))
            att = sa._attend(cast(res), B, T, tgt_mask)
            lin(att, sa.out_proj.weight, sa.out_proj.bias, epilogue=ops.EPI_RESID, resid=res)
            x1 = self._n(self.norm1, res, stage_embedding, torch.float32)
            catt = self._cross(cast(x1), mem_kv, B, T, S)
            lin(catt, ca.out_proj.weight, ca.out_proj.bias, epilogue=ops.EPI_RESID, resid=x1)
            x2 = self._n(self.norm2, x1, stage_embedding, torch.float32)
            h = lin(cast(x2), self.linear1.weight, self.linear1.bias, epilogue=ops.EPI_RELU)
            lin(h, self.linear2.weight, self.linear2.bias, epilogue=ops.EPI_RESID, resid=x2)
            out = self._n(self.norm3, x2, stage_embedding, torch.float32)
        out = out.view(B, T, d)
        if not self.batch_first:
            out = out.transpose(0, 1)
        return (out, stage_embedding) if tgt_is_tuple else out


class TransformerDecoder(_HipModule):
    """The container VALL-F builds its decoders with -- ``nn.TransformerDecoder`` of torch 1.13 (the reference's pin,
    README.md:31; valle/models/valle.py:141-152): N deep-copied layers, each called with the previous output and the memory,
    then the optional final norm.  Same attribute names (``layers``, ``num_layers``, ``norm``), hence the same state-dict keys."""

    def __init__(self, decoder_layer, num_layers, norm=None):
        super().__init__()
        self.layers = _get_clones(decoder_layer, num_layers)
        self.num_layers = num_layers
        self.norm = norm

    def forward(self, tgt, memory: Tensor, tgt_mask: Optional[Tensor] = None, memory_mask: Optional[Tensor] = None,
                tgt_key_padding_mask: Optional[Tensor] = None, memory_key_padding_mask: Optional[Tensor] = None):
        output = tgt
        for mod in self.layers:
            output = mod(output, memory, tgt_mask=tgt_mask, memory_mask=memory_mask, tgt_key_padding_mask=tgt_key_padding_mask,
                         memory_key_padding_mask=memory_key_padding_mask)
        if self.norm is not None:
            output = self.norm(output)
        return output
