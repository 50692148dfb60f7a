"""Tensor-level wrappers of the stand-alone HIP operators (``vle_op_*`` in valle_engine.h).

These are the kernels behind the Transformer block modules (``modules.py``); they exist as an
API so each kernel can be checked against its PyTorch fp32 definition in isolation.
All tensors must live on a ROCm device; nothing here computes on the host.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib

EPI_STORE, EPI_RELU, EPI_RESID, EPI_F32 = 0, 1, 2, 3


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return _lib.DTYPE_F32
    if t.dtype == torch.bfloat16:
        return _lib.DTYPE_BF16
    raise TypeError(f"unsupported dtype {t.dtype}")


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _st(t: torch.Tensor):
    assert t.is_cuda, "HIP operators need device tensors (no CPU fallback)"
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, out_dtype=torch.float32) -> torch.Tensor:
    """F.layer_norm(x, (d,), gamma, beta, 1e-5) with fp32 input and fp32/bf16 output."""
    lib = _lib.load()
    x = x.contiguous()
    assert x.dtype == torch.float32 and gamma.dtype == torch.float32 and beta.dtype == torch.float32
    d = x.shape[-1]
    rows = x.numel() // d
    out = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    _lib.check(lib.vle_op_layernorm(_st(x), _dt(out), _p(x), _p(gamma.contiguous()), _p(beta.contiguous()), _p(out), rows, d))
    return out


_WORKSPACES = {}


def linear_workspace(device: torch.device) -> torch.Tensor:
    """Per-device split-K workspace of the M <= 64 GEMM (tickets zeroed once; calls must share one stream)."""
    key = (device.type, device.index or 0)
    ws = _WORKSPACES.get(key)
    if ws is None:
        ws = torch.zeros(int(_lib.load().vle_op_linear_workspace_bytes()), dtype=torch.uint8, device=device)
        _WORKSPACES[key] = ws
    return ws


def linear(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, epilogue: int = EPI_STORE,
           resid: Optional[torch.Tensor] = None, ksplit: Optional[int] = 0) -> torch.Tensor:
    """epilogue(a @ w.T + bias) on the MFMA GEMM path. a (M,K), w (N,K) same dtype (fp32|bf16).
    ksplit (bf16, M <= 64 only): 0 = split K across workgroups as the engine does, n = force n slices,
    None = never split (vle_op_linear without a workspace)."""
    lib = _lib.load()
    a, w = a.contiguous(), w.contiguous()
    assert a.dtype == w.dtype and a.dim() == 2 and w.dim() == 2 and a.shape[1] == w.shape[1]
    M, K = a.shape
    N = w.shape[0]
    out = None
    if epilogue == EPI_RESID:
        assert resid is not None and resid.dtype == torch.float32 and resid.is_contiguous() and resid.shape == (M, N)
    elif epilogue == EPI_F32:
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    else:
        out = torch.empty(M, N, dtype=a.dtype, device=a.device)
    b = None if bias is None else bias.contiguous()
    if ksplit is None:
        _lib.check(lib.vle_op_linear(_st(a), _dt(a), _p(a), _p(w), _p(b), _p(out), _p(resid), M, N, K, epilogue))
    else:
        ws = linear_workspace(a.device)
        _lib.check(lib.vle_op_linear_ws(_st(a), _dt(a), _p(a), _p(w), _p(b), _p(out), _p(resid), M, N, K, epilogue, _p(ws), int(ksplit)))
    return resid if epilogue == EPI_RESID else out


def linear_ln_producer(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], resid: torch.Tensor, gamma: torch.Tensor):
    """resid (M,N) fp32 += a @ w.T + bias, in place; returns (xg bf16 (M,N) = resid * gamma, stats fp32 (N/64, M, 2) = (mean, M2) per
    64-column group, group-major) -- the producer half of the LayerNorm folded into the packed-row GEMMs (vle_op_linear_ln_producer)."""
    lib = _lib.load()
    a, w = a.contiguous(), w.contiguous()
    assert a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and resid.dtype == torch.float32 and resid.is_contiguous()
    M, K = a.shape
    N = w.shape[0]
    xg = torch.empty(M, N, dtype=torch.bfloat16, device=a.device)
    stats = torch.empty(N // 64, M, 2, dtype=torch.float32, device=a.device)
    b = None if bias is None else bias.contiguous()
    _lib.check(lib.vle_op_linear_ln_producer(_st(a), _p(a), _p(w), _p(b), _p(resid), _p(gamma.contiguous()), _p(xg), _p(stats), M, N, K))
    return xg, stats


def linear_ln_consumer(xg: torch.Tensor, w: torch.Tensor, tb: torch.Tensor, sg: torch.Tensor, stats: torch.Tensor, relu: bool = False) -> torch.Tensor:
    """bf16(act(rstd * (xg @ w.T - mean * sg) + tb)) with the rows' statistics combined from `stats` (vle_op_linear_ln_consumer)."""
    lib = _lib.load()
    xg, w = xg.contiguous(), w.contiguous()
    assert xg.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and stats.dtype == torch.float32 and stats.is_contiguous()
    M, K = xg.shape
    N = w.shape[0]
    assert stats.shape == (K // 64, M, 2)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=xg.device)
    _lib.check(lib.vle_op_linear_ln_consumer(_st(xg), _p(xg), _p(w), _p(tb.contiguous()), _p(sg.contiguous()), _p(stats), _p(out), M, N, K, int(relu)))
    return out


def tune(name: str, value: int) -> None:
    """Process-global kernel-selection knob of the stand-alone operators (vle_op_tune)."""
    _lib.check(_lib.load().vle_op_tune(name.encode(), int(value)))


def linear_skinny(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, epilogue: int = 0,
                  resid: Optional[torch.Tensor] = None, gamma: Optional[torch.Tensor] = None,
                  beta: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Weight-streaming path of the AR step: x fp32 (M<=8, K), w (N,K) fp32|bf16; optional fused LayerNorm."""
    lib = _lib.load()
    x, w = x.contiguous(), w.contiguous()
    assert x.dtype == torch.float32
    M, K = x.shape
    N = w.shape[0]
    out = None if epilogue == 2 else torch.empty(M, N, dtype=torch.float32, device=x.device)
    b = None if bias is None else bias.contiguous()
    _lib.check(lib.vle_op_linear_skinny(_st(x), _dt(w), _p(x), _p(gamma), _p(beta), _p(w), _p(b), _p(out), _p(resid), M, N, K, epilogue))
    return resid if epilogue == 2 else out


def attention(qkv: torch.Tensor, seq_off: torch.Tensor, text_len: torch.Tensor, nhead: int, causal: bool) -> torch.Tensor:
    """Packed-sequence attention. qkv (rows, 3d); seq_off int32 (B+1,), text_len int32 (B,) on device."""
    lib = _lib.load()
    qkv = qkv.contiguous()
    rows, d3 = qkv.shape
    d = d3 // 3
    B = seq_off.numel() - 1
    so = seq_off.to(torch.int32).contiguous()
    tl = text_len.to(torch.int32).contiguous()
    lens = (so[1:] - so[:-1]).cpu()
    out = torch.empty(rows, d, dtype=qkv.dtype, device=qkv.device)
    _lib.check(lib.vle_op_attention(_st(qkv), _dt(qkv), _p(qkv), _p(out), _p(so), _p(tl), B, int(lens.max()), d, nhead, int(causal)))
    return out


def cross_attention(q: torch.Tensor, kv: torch.Tensor, nhead: int) -> torch.Tensor:
    """softmax(q K^T / sqrt(dh)) V per head, no mask: q (Tq, d), kv (S, 2d) = [K | V] of the memory sequence (VALL-F's
    ``multihead_attn``, valle/modules/transformer.py:582-597); fp32 or bf16, output like q."""
    lib = _lib.load()
    q, kv = q.contiguous(), kv.contiguous()
    assert q.dim() == 2 and kv.dim() == 2 and kv.shape[1] == 2 * q.shape[1] and q.dtype == kv.dtype
    out = torch.empty_like(q)
    _lib.check(lib.vle_op_cross_attention(_st(q), _dt(q), _p(q), _p(kv), _p(out), q.shape[0], kv.shape[0], q.shape[1], int(nhead)))
    return out


def decode_attention(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, kv_len: torch.Tensor, nsplit: int = 1,
                     merged: bool = True):
    """One new query per utterance against the head-major KV cache.
    q fp32 (B, H*dh); k_cache/v_cache (B, H, ctx_max, dh) fp32|bf16; kv_len int32 (B,) = slot of the newest key.
    Returns (out fp32 (B, d) or None, workspace) -- the workspace holds the split partials for attn_out_proj."""
    lib = _lib.load()
    q, k_cache, v_cache = q.contiguous(), k_cache.contiguous(), v_cache.contiguous()
    assert q.dtype == torch.float32 and k_cache.dtype == v_cache.dtype and k_cache.shape == v_cache.shape
    B, H, ctx_max, dh = k_cache.shape
    d = H * dh
    ws = torch.empty(B * nsplit * (d + 2 * H), dtype=torch.float32, device=q.device)
    out = torch.empty(B, d, dtype=torch.float32, device=q.device) if merged else None
    kl = kv_len.to(torch.int32).contiguous()
    _lib.check(lib.vle_op_decode_attention(_st(q), _dt(k_cache), _p(q), _p(k_cache), _p(v_cache), _p(kl), _p(ws), _p(out), B, H, dh,
                                           ctx_max, nsplit))
    return out, ws


def attn_step1(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, w_in: torch.Tensor, b_in: torch.Tensor, w_out: torch.Tensor,
               b_out: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, kv_len: int, nhead: int, nsplit: int = 8) -> torch.Tensor:
    """The attention half of one utterance's decode step as the batch-1 AR step runs it (vle_op_attn_step1): returns
    x + out_proj(MHA(LayerNorm(x))) for the new token (x fp32 (d,)), appends its K / V to slot ``kv_len`` of the caches
    (nhead, ctx_max, dh) IN PLACE and attends slots 0 .. kv_len.  w_in (3d, d), w_out (d, d) fp32|bf16, caches of the same dtype."""
    lib = _lib.load()
    assert x.dtype == torch.float32 and x.dim() == 1 and w_in.dtype == w_out.dtype == k_cache.dtype == v_cache.dtype
    assert k_cache.is_contiguous() and v_cache.is_contiguous() and k_cache.shape == v_cache.shape and k_cache.shape[0] == nhead
    d = x.shape[0]
    _, ctx_max, dh = k_cache.shape
    assert nhead * dh == d and 0 <= kv_len < ctx_max
    x = x.clone().contiguous()
    w_in, w_out = w_in.contiguous(), w_out.contiguous()
    ws = torch.empty(3 * d + nsplit * (d + 2 * nhead), dtype=torch.float32, device=x.device)
    kl = torch.tensor([kv_len], dtype=torch.int32, device=x.device)
    _lib.check(lib.vle_op_attn_step1(_st(x), _dt(w_in), _p(x), _p(gamma.contiguous()), _p(beta.contiguous()), _p(w_in), _p(b_in.contiguous()),
                                     _p(w_out), _p(b_out.contiguous()), _p(k_cache), _p(v_cache), _p(kl), _p(ws), nhead, dh, ctx_max, nsplit))
    return x


def attn_out_proj(ws: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], resid: torch.Tensor, nhead: int,
                  nsplit: int) -> torch.Tensor:
    """resid (B, d) fp32 += merge(split partials in ws) @ w.T + bias  -- out_proj of the AR step."""
    lib = _lib.load()
    w = w.contiguous()
    B, d = resid.shape
    assert resid.dtype == torch.float32 and resid.is_contiguous() and w.shape == (d, d)
    b = None if bias is None else bias.contiguous()
    _lib.check(lib.vle_op_attn_out_proj(_st(resid), _dt(w), _p(ws), _p(w), _p(b), _p(resid), B, nhead, d // nhead, nsplit))
    return resid


def token_embedding(ids: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
    """TokenEmbedding.forward (valle/modules/embedding.py:43-47): table[ids]; ids int64 (...), table fp32 (V, d)."""
    lib = _lib.load()
    assert ids.dtype == torch.int64 and table.dtype == torch.float32 and table.dim() == 2
    ids, table = ids.contiguous(), table.contiguous()
    d = table.shape[1]
    if ids.numel():  # nn.Embedding raises IndexError (valle/modules/embedding.py:34,44); the kernel gathers table + id * d unchecked
        lo, hi = torch.aminmax(ids)
        if int(lo) < 0 or int(hi) >= table.shape[0]:
            raise IndexError(f"token id out of range for an embedding table of {table.shape[0]} rows: min {int(lo)}, max {int(hi)}")
    out = torch.empty(*ids.shape, d, dtype=torch.float32, device=table.device)
    _lib.check(lib.vle_op_token_embedding(_st(table), _p(ids), _p(table), _p(out), ids.numel(), d))
    return out


def token_embedding_add(inout: torch.Tensor, ids: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
    """``inout += table[ids]`` in place (``y_emb[:, :P] += nar_audio_embeddings[j](codes)``, valle.py:1104-1113, 1134): inout fp32
    (..., d) CONTIGUOUS (a view of whole rows is fine), ids int64 with one id per row."""
    lib = _lib.load()
    assert inout.dtype == torch.float32 and inout.is_contiguous() and ids.dtype == torch.int64 and table.dtype == torch.float32
    ids, table = ids.contiguous(), table.contiguous()
    d = table.shape[1]
    assert inout.shape[-1] == d and inout.numel() == ids.numel() * d
    if ids.numel():
        lo, hi = torch.aminmax(ids)
        if int(lo) < 0 or int(hi) >= table.shape[0]:
            raise IndexError(f"token id out of range for an embedding table of {table.shape[0]} rows: min {int(lo)}, max {int(hi)}")
    _lib.check(lib.vle_op_token_embedding_add(_st(table), _p(ids), _p(table), _p(inout), ids.numel(), d))
    return inout


def sine_positional(x: torch.Tensor, pe: torch.Tensor, alpha: torch.Tensor, x_scale: float = 1.0) -> torch.Tensor:
    """SinePositionalEmbedding.forward (embedding.py:93-97): x * x_scale + alpha * pe[:T]; x fp32 (B, T, d), pe (>=T, d)."""
    lib = _lib.load()
    assert x.dtype == torch.float32 and x.dim() == 3 and pe.dtype == torch.float32 and alpha.dtype == torch.float32
    x, pe = x.contiguous(), pe.contiguous()
    B, T, d = x.shape
    assert pe.shape[-1] == d and pe.numel() // d >= T
    out = torch.empty_like(x)
    _lib.check(lib.vle_op_sine_positional(_st(x), _p(x), _p(pe), _p(alpha), float(x_scale), _p(out), B, T, d))
    return out


def adaln_fold(wb: torch.Tensor, g: torch.Tensor, be: torch.Tensor):
    """[w ; b] = project_layer(stage_emb) (2d,) with the inner LayerNorm affine -> (w * g, w * be + b)
    (AdaptiveLayerNorm.forward, transformer.py:93-108, as the affine of one LayerNorm call)."""
    lib = _lib.load()
    d = g.numel()
    assert wb.dtype == g.dtype == be.dtype == torch.float32 and wb.numel() == 2 * d and be.numel() == d
    out = torch.empty(2, d, dtype=torch.float32, device=g.device)
    _lib.check(lib.vle_op_adaln_fold(_st(g), _p(wb.contiguous()), _p(g.contiguous()), _p(be.contiguous()), _p(out[0]), _p(out[1]), d))
    return out[0], out[1]


# ---- FP8W (VLE_DTYPE_FP8W): fp8 e4m3fn weight rows with one power-of-two scale each ------------------------
def quantize_fp8w(w: torch.Tensor):
    """The engine's host-side weight quantiser (vle_quantize_fp8w): w fp32 (N, K) on the CPU ->
    (q uint8 (N, K) e4m3fn codes, scale fp32 (N,), deq fp32 (N, K) = W').  No GPU involved."""
    lib = _lib.load()
    w = w.detach().to("cpu", torch.float32).contiguous()
    assert w.dim() == 2
    N, K = w.shape
    q = torch.empty(N, K, dtype=torch.uint8)
    scale = torch.empty(N, dtype=torch.float32)
    deq = torch.empty(N, K, dtype=torch.float32)
    _lib.check(lib.vle_quantize_fp8w(C.c_void_p(w.data_ptr()), N, K, C.c_void_p(q.data_ptr()), C.c_void_p(scale.data_ptr()),
                                     C.c_void_p(deq.data_ptr())))
    return q, scale, deq


def linear_skinny_fp8w(x: torch.Tensor, w8: torch.Tensor, wscale: torch.Tensor, bias: Optional[torch.Tensor] = None, epilogue: int = 0,
                       resid: Optional[torch.Tensor] = None, gamma: Optional[torch.Tensor] = None,
                       beta: Optional[torch.Tensor] = None) -> torch.Tensor:
    """linear_skinny for ONE row on FP8W weights (the batch-1 AR step): x fp32 (1, K), w8 uint8 (N, K), wscale fp32 (N,)."""
    lib = _lib.load()
    x, w8, wscale = x.contiguous(), w8.contiguous(), wscale.contiguous()
    assert x.dtype == torch.float32 and w8.dtype == torch.uint8 and wscale.dtype == torch.float32 and x.shape[0] == 1
    K = x.shape[1]
    N = w8.shape[0]
    out = None if epilogue == 2 else torch.empty(1, N, dtype=torch.float32, device=x.device)
    b = None if bias is None else bias.contiguous()
    _lib.check(lib.vle_op_linear_skinny_fp8w(_st(x), _p(x), _p(gamma), _p(beta), _p(w8), _p(wscale), _p(b), _p(out), _p(resid), N, K, epilogue))
    return resid if epilogue == 2 else out


def linear_fp8w(a: torch.Tensor, w8: torch.Tensor, wscale: torch.Tensor, bias: Optional[torch.Tensor] = None, epilogue: int = EPI_STORE,
                resid: Optional[torch.Tensor] = None, ksplit: int = 0) -> torch.Tensor:
    """linear for M <= 64 bf16 rows on FP8W weights (the AR step of 2..64 utterances)."""
    lib = _lib.load()
    a, w8, wscale = a.contiguous(), w8.contiguous(), wscale.contiguous()
    assert a.dtype == torch.bfloat16 and w8.dtype == torch.uint8 and wscale.dtype == torch.float32
    M, K = a.shape
    N = w8.shape[0]
    out = None
    if epilogue == EPI_RESID:
        assert resid is not None and resid.dtype == torch.float32 and resid.is_contiguous() and resid.shape == (M, N)
    elif epilogue == EPI_F32:
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    else:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=a.device)
    b = None if bias is None else bias.contiguous()
    ws = linear_workspace(a.device)
    _lib.check(lib.vle_op_linear_fp8w(_st(a), _p(a), _p(w8), _p(wscale), _p(b), _p(out), _p(resid), M, N, K, epilogue, _p(ws), int(ksplit)))
    return resid if epilogue == EPI_RESID else out


def cross_entropy_rows(logits: torch.Tensor, targets: torch.Tensor, ignore_index: int = -100, topk: int = 10):
    """Per-row cross-entropy and top-k hit flags of the teacher-forced forward (valle.py:875, 877-879, 936-956).
    logits fp32 (rows, V), targets int64 (rows,) -> (loss fp32 (rows,), hit int32 (rows,): 1 / 0, -1 = ignored)."""
    lib = _lib.load()
    assert logits.dtype == torch.float32 and logits.dim() == 2 and targets.dtype == torch.int64 and targets.shape == (logits.shape[0],)
    logits, targets = logits.contiguous(), targets.contiguous()
    loss = torch.empty(logits.shape[0], dtype=torch.float32, device=logits.device)
    hit = torch.empty(logits.shape[0], dtype=torch.int32, device=logits.device)
    _lib.check(lib.vle_op_cross_entropy(_st(logits), _p(logits), _p(targets), _p(loss), _p(hit), logits.shape[0], logits.shape[1],
                                        int(ignore_index), int(topk)))
    return loss, hit


def topk_sample(logits: torch.Tensor, top_k: int = -100, temperature: float = 1.0, seed: int = 0, step: int = 0):
    """``topk_sampling(logits, top_k, top_p=1.0, temperature)`` (valle/models/valle.py:1287-1302) per row of fp32 ``logits``
    (rows, V): returns (samples, argmax), both int64 (rows,).  Row r draws from the RNG stream of request r of seed ``seed``
    at AR step ``step`` (the engine's own streams: a block-level decode loop samples what ``vle_ar_generate`` would)."""
    lib = _lib.load()
    assert logits.dim() == 2 and logits.dtype == torch.float32
    logits = logits.contiguous()
    rows, V = logits.shape
    samples = torch.empty(rows, dtype=torch.int64, device=logits.device)
    argmax = torch.empty(rows, dtype=torch.int64, device=logits.device)
    _lib.check(lib.vle_op_topk_sample(_st(logits), _p(logits), rows, V, int(top_k), float(temperature), int(seed) & (2**64 - 1),
                                      int(step) & 0xFFFFFFFF, _p(samples), _p(argmax)))
    return samples, argmax


def quantize_rows_fp8(x: torch.Tensor):
    """Per-row activation quantiser of engine mode FP8: x bf16 (rows, K) -> (codes uint8 (rows, K) e4m3fn, scale fp32 (rows,))."""
    lib = _lib.load()
    assert x.dtype == torch.bfloat16 and x.dim() == 2
    x = x.contiguous()
    q = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    sc = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    _lib.check(lib.vle_op_quantize_rows_fp8(_st(x), _p(x), _p(q), _p(sc), x.shape[0], x.shape[1]))
    return q, sc


def linear_fp8(a8: torch.Tensor, a_scale: torch.Tensor, w8: torch.Tensor, w_scale: torch.Tensor, bias: Optional[torch.Tensor] = None,
               epilogue: int = EPI_STORE, resid: Optional[torch.Tensor] = None) -> torch.Tensor:
    """epilogue((a8 @ w8.T) * a_scale[:, None] * w_scale[None, :] + bias) on the block-scaled fp8 MFMA; codes uint8 e4m3fn."""
    lib = _lib.load()
    assert a8.dtype == torch.uint8 and w8.dtype == torch.uint8 and a8.shape[1] == w8.shape[1]
    a8, w8 = a8.contiguous(), w8.contiguous()
    M, K = a8.shape
    N = w8.shape[0]
    out = None
    if epilogue == EPI_RESID:
        assert resid is not None and resid.dtype == torch.float32 and resid.is_contiguous() and resid.shape == (M, N)
    elif epilogue == EPI_F32:
        out = torch.empty(M, N, dtype=torch.float32, device=a8.device)
    else:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=a8.device)
    b = None if bias is None else bias.contiguous()
    _lib.check(lib.vle_op_linear_fp8(_st(a8), _p(a8), _p(a_scale.contiguous()), _p(w8), _p(w_scale.contiguous()), _p(b), _p(out), _p(resid),
                                     M, N, K, epilogue))
    return resid if epilogue == EPI_RESID else out
