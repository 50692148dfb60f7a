"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the VALL-E AR+NAR decode path.

A functional (state_dict in, codes out) restatement, in plain fp32 torch-CPU tensor ops,
of the algorithm the reference executes in ``VALLE.inference()`` / ``VALLE.continual()``.
Every function cites the reference file:line it follows (paths relative to
/root/reference).  It is the *checker* for the HIP engine: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it; the
product path (``valle_amd/``) never does.

Parity status: PINNED.  ``oracle/make_golden.py`` runs the unmodified reference (imported
through ``oracle/ref_import.py``) on the weights produced by ``make_state_dict`` here and
commits the reference's outputs under ``tests/golden/``; ``tests/test_oracle_golden.py``
checks this restatement against those vectors (token ids exact, logits to fp32 noise).
The reference's own tests pin nothing numerically.

Deliberately *literal*: like the reference there is no KV cache -- every AR step re-runs
the whole [text; audio] sequence (valle/models/valle.py:1012-1057).  ``kv_cache=True``
selects an incremental evaluation that is exact by the prefix-LM mask argument
(valle.py:1019-1030) and is itself pinned against the literal path in the tests; it
exists so GPU parity tests can use sizes the literal path cannot finish in seconds.
"""
from __future__ import annotations

import hashlib
import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

NUM_TEXT_TOKENS = 512  # valle/models/macros.py:2
NUM_AUDIO_TOKENS = 1024  # valle/models/macros.py:5
LN_EPS = 1e-5  # valle/modules/transformer.py:27


@dataclass
class OracleConfig:
    """Constructor surface of VALLE (valle/models/valle.py:727-760, 54-84)."""

    d_model: int = 1024
    nhead: int = 16
    num_layers: int = 12
    norm_first: bool = True
    add_prenet: bool = False
    prefix_mode: int = 0
    share_embedding: bool = True
    nar_scale_factor: float = 1.0
    prepend_bos: bool = False
    num_quantizers: int = 8
    model: str = "valle"  # "valle" (decoder-only: text is a prefix of the sequence) | "vallf" (text is cross-attention memory, valle.py:50-710)

    # NAR sizes, valle.py:83, :235, :241
    @property
    def nar_d_model(self) -> int:
        return int(self.d_model * self.nar_scale_factor)

    @property
    def nar_nhead(self) -> int:
        return int(self.nhead * self.nar_scale_factor)

    @property
    def nar_num_layers(self) -> int:
        return int(self.num_layers * self.nar_scale_factor)

    @property
    def fused_shape(self) -> bool:
        """The production shape the fused HIP engine runs (pre-norm, no prenet, one width); the other constructor
        combinations decode through the block modules (valle_amd/model.py)."""
        return self.model == "valle" and self.norm_first and not self.add_prenet and self.nar_scale_factor == 1.0

    def check_supported(self):
        assert self.model in ("valle", "vallf")
        assert self.nar_d_model > 0 and self.nar_nhead > 0 and self.nar_num_layers > 0 and self.nar_d_model % self.nar_nhead == 0


# --------------------------------------------------------------------------------------
# deterministic weights (same on every machine with this torch build; no reference needed)
# --------------------------------------------------------------------------------------
def state_dict_spec(cfg: OracleConfig) -> "OrderedDict[str, tuple]":
    """Key -> shape, exactly the reference's ``state_dict()`` (valle.py:85-279)."""
    d, L, Q = cfg.d_model, cfg.num_layers, cfg.num_quantizers
    nd, nL = cfg.nar_d_model, cfg.nar_num_layers
    s: "OrderedDict[str, tuple]" = OrderedDict()
    s["ar_text_embedding.word_embeddings.weight"] = (NUM_TEXT_TOKENS, d)
    s["nar_text_embedding.word_embeddings.weight"] = (NUM_TEXT_TOKENS, nd)
    s["ar_audio_embedding.word_embeddings.weight"] = (NUM_AUDIO_TOKENS + 1 + int(cfg.prepend_bos), d)

    def prenets(prefix, w):  # valle.py:100-125 / :183-216: nn.Sequential indices of the modules that own parameters
        for conv, bn in ((1, 2), (5, 6), (9, 10)):
            s[f"{prefix}_text_prenet.{conv}.weight"] = (w, w, 5)
            s[f"{prefix}_text_prenet.{conv}.bias"] = (w,)
            s[f"{prefix}_text_prenet.{bn}.weight"] = (w,)
            s[f"{prefix}_text_prenet.{bn}.bias"] = (w,)
            s[f"{prefix}_text_prenet.{bn}.running_mean"] = (w,)
            s[f"{prefix}_text_prenet.{bn}.running_var"] = (w,)
            s[f"{prefix}_text_prenet.{bn}.num_batches_tracked"] = ()
        s[f"{prefix}_text_prenet.14.weight"] = (w, w)
        s[f"{prefix}_text_prenet.14.bias"] = (w,)
        s[f"{prefix}_audio_prenet.0.weight"] = (256, w)
        s[f"{prefix}_audio_prenet.0.bias"] = (256,)
        s[f"{prefix}_audio_prenet.3.weight"] = (256, 256)
        s[f"{prefix}_audio_prenet.3.bias"] = (256,)
        s[f"{prefix}_audio_prenet.6.weight"] = (w, 256)
        s[f"{prefix}_audio_prenet.6.bias"] = (w,)

    if cfg.add_prenet:
        prenets("ar", d)
    s["ar_text_position.alpha"] = (1,)
    s["ar_audio_position.alpha"] = (1,)

    def layer(prefix, adaptive, d):
        s[f"{prefix}.self_attn.in_proj_weight"] = (3 * d, d)
        s[f"{prefix}.self_attn.in_proj_bias"] = (3 * d,)
        s[f"{prefix}.self_attn.out_proj.weight"] = (d, d)
        s[f"{prefix}.self_attn.out_proj.bias"] = (d,)
        if cfg.model == "vallf":  # TransformerDecoderLayer, transformer.py:442-450
            s[f"{prefix}.multihead_attn.in_proj_weight"] = (3 * d, d)
            s[f"{prefix}.multihead_attn.in_proj_bias"] = (3 * d,)
            s[f"{prefix}.multihead_attn.out_proj.weight"] = (d, d)
            s[f"{prefix}.multihead_attn.out_proj.bias"] = (d,)
        s[f"{prefix}.linear1.weight"] = (4 * d, d)
        s[f"{prefix}.linear1.bias"] = (4 * d,)
        s[f"{prefix}.linear2.weight"] = (d, 4 * d)
        s[f"{prefix}.linear2.bias"] = (d,)
        for n in ("norm1", "norm2", "norm3") if cfg.model == "vallf" else ("norm1", "norm2"):
            if adaptive:
                s[f"{prefix}.{n}.project_layer.weight"] = (2 * d, d)
                s[f"{prefix}.{n}.project_layer.bias"] = (2 * d,)
                s[f"{prefix}.{n}.norm.weight"] = (d,)
                s[f"{prefix}.{n}.norm.bias"] = (d,)
            else:
                s[f"{prefix}.{n}.weight"] = (d,)
                s[f"{prefix}.{n}.bias"] = (d,)

    for l in range(L):
        layer(f"ar_decoder.layers.{l}", False, d)
    if cfg.norm_first:  # norm=LayerNorm(d) if norm_first else None, valle.py:151
        s["ar_decoder.norm.weight"] = (d,)
        s["ar_decoder.norm.bias"] = (d,)
    s["ar_predict_layer.weight"] = (NUM_AUDIO_TOKENS + 1, d)
    if Q > 1:
        s["nar_audio_embeddings.0.word_embeddings.weight"] = (NUM_AUDIO_TOKENS + 1, nd)
        for j in range(1, Q):
            s[f"nar_audio_embeddings.{j}.word_embeddings.weight"] = (NUM_AUDIO_TOKENS, nd)
        if cfg.add_prenet:
            prenets("nar", nd)
        s["nar_text_position.alpha"] = (1,)
        s["nar_audio_position.alpha"] = (1,)
        for l in range(nL):
            layer(f"nar_decoder.layers.{l}", True, nd)
        if cfg.norm_first:  # valle.py:242-246
            s["nar_decoder.norm.project_layer.weight"] = (2 * nd, nd)
            s["nar_decoder.norm.project_layer.bias"] = (2 * nd,)
            s["nar_decoder.norm.norm.weight"] = (nd,)
            s["nar_decoder.norm.norm.bias"] = (nd,)
        for i in range(Q - 1):
            s[f"nar_predict_layers.{i}.weight"] = (NUM_AUDIO_TOKENS, nd)
        for i in range(Q - 1):
            s[f"nar_stage_embeddings.{i}.word_embeddings.weight"] = (1, nd)
    return s


def _key_seed(seed: int, key: str) -> int:
    h = hashlib.sha256(f"{seed}:{key}".encode()).digest()
    return int.from_bytes(h[:7], "little")


def make_state_dict(cfg: OracleConfig, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """Deterministic synthetic weights with the reference's init *distributions*
    (xavier-uniform in_proj: valle/modules/activation.py:175-190; nn.Linear / nn.Embedding
    defaults elsewhere), but every layer distinct and biases / LN affine / alpha perturbed so
    that every term of the computation is exercised.  Weight tying as valle.py:261-271."""
    cfg.check_supported()
    spec = state_dict_spec(cfg)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for key, shape in spec.items():
        g = torch.Generator().manual_seed(_key_seed(seed, key))
        if key.endswith("num_batches_tracked"):  # BatchNorm1d buffer (int64 scalar; unused in eval mode)
            sd[key] = torch.tensor(100, dtype=torch.int64)
            continue
        if key.endswith("running_mean"):
            t = 0.1 * torch.randn(shape, generator=g)
        elif key.endswith("running_var"):
            t = 0.5 + torch.rand(shape, generator=g)
        elif "_prenet." in key and len(shape) == 1 and key.endswith("weight"):  # BatchNorm1d affine
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif "_prenet." in key and len(shape) == 3:  # Conv1d (out, in, k): nn.Conv1d's default bound 1 / sqrt(in * k)
            a = 1.0 / math.sqrt(shape[1] * shape[2])
            t = (torch.rand(shape, generator=g) * 2 - 1) * a
        elif key.endswith("alpha"):
            t = 1.0 + 0.25 * torch.rand(shape, generator=g) if key.startswith("ar_") else torch.ones(shape)
        elif "word_embeddings" in key:
            t = torch.randn(shape, generator=g)
        elif key.endswith("in_proj_weight"):
            a = math.sqrt(6.0 / (shape[0] + shape[1]))
            t = (torch.rand(shape, generator=g) * 2 - 1) * a
        elif ".norm" in key and "project_layer" not in key and key.endswith("weight") and len(shape) == 1:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif ".norm" in key and "project_layer" not in key and key.endswith("bias"):
            t = 0.05 * torch.randn(shape, generator=g)
        elif key.endswith("weight"):
            a = 1.0 / math.sqrt(shape[1])
            t = (torch.rand(shape, generator=g) * 2 - 1) * a
        elif key.endswith("bias"):
            t = (torch.rand(shape, generator=g) * 2 - 1) * 0.02
        else:
            raise KeyError(key)
        sd[key] = t.to(torch.float32).contiguous()
    if cfg.share_embedding and cfg.num_quantizers > 1:
        for j in range(0, cfg.num_quantizers - 2):  # valle.py:268-271
            sd[f"nar_predict_layers.{j}.weight"] = sd[f"nar_audio_embeddings.{j + 2}.word_embeddings.weight"]
    return sd


def make_inputs(S: int, P: int, seed: int = 1234, Q: int = 8):
    """Synthetic inputs of SURVEY.md 8(d): text ids uniform in [3,100) with BOS=1 / EOS=2
    (valle/data/collation.py:49-57), prompt codes uniform in [0,1024)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randint(3, 100, (1, S), generator=g, dtype=torch.int64)
    x[0, 0] = 1
    x[0, -1] = 2
    x_lens = torch.tensor([S], dtype=torch.int32)
    y = torch.randint(0, NUM_AUDIO_TOKENS, (1, P, 8), generator=g, dtype=torch.int64)
    return x, x_lens, y[..., :Q].contiguous()


# --------------------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------------------
def sine_pe(T: int, d: int) -> torch.Tensor:
    """valle/modules/embedding.py:75-91 (reverse=False); returns (T, d) fp32."""
    pe = torch.zeros(T, d)
    position = torch.arange(0, T, dtype=torch.float32).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def sine_position(x: torch.Tensor, alpha: torch.Tensor, start: int = 0) -> torch.Tensor:
    """SinePositionalEmbedding.forward, embedding.py:93-97 with x_scale = 1 (scale=False at
    valle.py:131,137,221,227) and dropout = identity in eval.  ``start`` only for kv_cache mode."""
    T, d = x.shape[-2], x.shape[-1]
    pe = sine_pe(start + T, d)[start:].to(x.device)  # the table is built on the host like the reference's (embedding.py:75-91)
    return x * 1.0 + alpha * pe


def token_embedding(sd, name: str, ids: torch.Tensor) -> torch.Tensor:
    """TokenEmbedding.forward, embedding.py:43-47 (dropout p=0)."""
    return F.embedding(ids, sd[f"{name}.word_embeddings.weight"])


def layer_norm(x, w, b):
    """LayerNorm.forward, valle/modules/transformer.py:57-74 -> F.layer_norm, eps 1e-5."""
    return F.layer_norm(x, (x.shape[-1],), w, b, LN_EPS)


def norm_site(sd, prefix: str, x, stage_emb):
    """Plain LayerNorm (AR) or AdaptiveLayerNorm (NAR): transformer.py:93-108 --
    [w, b] = split(project_layer(stage_emb)); w * norm(x) + b."""
    if stage_emb is None:
        return layer_norm(x, sd[f"{prefix}.weight"], sd[f"{prefix}.bias"])
    wb = F.linear(stage_emb, sd[f"{prefix}.project_layer.weight"], sd[f"{prefix}.project_layer.bias"])
    d = x.shape[-1]
    w, b = wb[..., :d], wb[..., d:]
    return w * layer_norm(x, sd[f"{prefix}.norm.weight"], sd[f"{prefix}.norm.bias"]) + b


def fp8a_rows(x: torch.Tensor) -> torch.Tensor:
    """Engine mode FP8 (valle_amd/csrc/misc.hip quantize_rows_fp8_kernel): the activations of a Linear in the packed passes are
    rounded to bf16, then per row to e4m3fn with one power-of-two scale (the rule of fp8w_quantize); returns the dequantised
    values the fp8 MFMA effectively multiplies."""
    xb = x.to(torch.bfloat16).to(torch.float32)
    return fp8w_quantize(xb)[2]


def _lin(x, w, b, act_fp8: bool):
    return F.linear(fp8a_rows(x) if act_fp8 else x, w, b)


def mha(sd, prefix: str, x, nhead: int, attn_mask: Optional[torch.Tensor], kv_state=None, act_fp8: bool = False):
    """MultiheadAttention.forward (valle/modules/activation.py:199-431) ->
    F.multi_head_attention_forward: packed in-proj rows [Q;K;V], heads = contiguous dh slices,
    softmax(Q K^T / sqrt(dh) + mask) V, out_proj.  ``attn_mask``: bool (T,T), True = blocked
    (valle.py:1019-1033).  x: (T, d)."""
    T, d = x.shape
    dh = d // nhead
    qkv = _lin(x, sd[f"{prefix}.in_proj_weight"], sd[f"{prefix}.in_proj_bias"], act_fp8)
    q, k, v = qkv[:, :d], qkv[:, d : 2 * d], qkv[:, 2 * d :]
    if kv_state is not None:  # incremental mode: append to cached keys/values
        if kv_state.get("k") is not None:
            k = torch.cat([kv_state["k"], k], 0)
            v = torch.cat([kv_state["v"], v], 0)
        kv_state["k"], kv_state["v"] = k, v
    Tk = k.shape[0]
    qh = q.view(T, nhead, dh).transpose(0, 1)  # (h, T, dh)
    kh = k.view(Tk, nhead, dh).transpose(0, 1)
    vh = v.view(Tk, nhead, dh).transpose(0, 1)
    scores = torch.matmul(qh, kh.transpose(1, 2)) / math.sqrt(dh)
    if attn_mask is not None:
        scores = scores.masked_fill(attn_mask[None], float("-inf"))
    p = torch.softmax(scores, dim=-1)
    o = torch.matmul(p, vh).transpose(0, 1).reshape(T, d)
    return _lin(o, sd[f"{prefix}.out_proj.weight"], sd[f"{prefix}.out_proj.bias"], act_fp8)


def encoder_layer(sd, prefix: str, x, nhead: int, attn_mask, stage_emb, kv_state=None, act_fp8: bool = False, norm_first: bool = True):
    """TransformerEncoderLayer.forward (transformer.py:296-308).  Pre-norm: x += SA(norm1(x)); x += W2 relu(W1 norm2(x) + b1) + b2;
    post-norm: x = norm1(x + SA(x)); x = norm2(x + FFN(x))  (ReLU: transformer.py:187, :333)."""
    if not norm_first:  # :303-308
        x = norm_site(sd, f"{prefix}.norm1", x + mha(sd, f"{prefix}.self_attn", x, nhead, attn_mask, kv_state, act_fp8), stage_emb)
        h = F.relu(_lin(x, sd[f"{prefix}.linear1.weight"], sd[f"{prefix}.linear1.bias"], act_fp8))
        return norm_site(sd, f"{prefix}.norm2", x + _lin(h, sd[f"{prefix}.linear2.weight"], sd[f"{prefix}.linear2.bias"], act_fp8), stage_emb)
    x = x + mha(sd, f"{prefix}.self_attn", norm_site(sd, f"{prefix}.norm1", x, stage_emb), nhead, attn_mask, kv_state, act_fp8)
    h = F.relu(_lin(norm_site(sd, f"{prefix}.norm2", x, stage_emb), sd[f"{prefix}.linear1.weight"], sd[f"{prefix}.linear1.bias"], act_fp8))
    x = x + _lin(h, sd[f"{prefix}.linear2.weight"], sd[f"{prefix}.linear2.bias"], act_fp8)
    return x


def encoder(sd, prefix: str, cfg: OracleConfig, x, attn_mask=None, stage_emb=None, kv_states=None, layer_states=None,
            act_fp8: bool = False):
    """TransformerEncoder.forward (transformer.py:363-406): the layers, then -- for norm_first models only -- the final norm
    (LayerNorm for AR, valle.py:151; AdaptiveLayerNorm(nn.LayerNorm) for NAR, valle.py:242-246).  The NAR decoder has
    int(L * scale) layers of int(h * scale) heads (valle.py:235, :241)."""
    nar = prefix.startswith("nar_")
    for l in range(cfg.nar_num_layers if nar else cfg.num_layers):
        x = encoder_layer(
            sd, f"{prefix}.layers.{l}", x, cfg.nar_nhead if nar else cfg.nhead, attn_mask, stage_emb,
            None if kv_states is None else kv_states[l], act_fp8, cfg.norm_first,
        )
        if layer_states is not None:
            layer_states.append(x.clone())
    return norm_site(sd, f"{prefix}.norm", x, stage_emb) if cfg.norm_first else x


def text_prenet(sd, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """{ar,nar}_text_prenet (valle.py:100-116): 3 x [Conv1d(k 5, padding "same") -> BatchNorm1d (eval: running statistics)
    -> ReLU -> Dropout (identity)] over time, then Linear.  x (T, d) -> (T, d)."""
    h = x.t()[None]  # Transpose(): (1, d, T)
    for conv, bn in ((1, 2), (5, 6), (9, 10)):
        h = F.conv1d(h, sd[f"{prefix}.{conv}.weight"], sd[f"{prefix}.{conv}.bias"], padding=2)
        h = F.batch_norm(h, sd[f"{prefix}.{bn}.running_mean"], sd[f"{prefix}.{bn}.running_var"], sd[f"{prefix}.{bn}.weight"],
                         sd[f"{prefix}.{bn}.bias"], False, 0.1, 1e-5)
        h = F.relu(h)
    return F.linear(h[0].t(), sd[f"{prefix}.14.weight"], sd[f"{prefix}.14.bias"])


def audio_prenet(sd, prefix: str, y: torch.Tensor) -> torch.Tensor:
    """{ar,nar}_audio_prenet (valle.py:118-126): Linear(d, 256) ReLU Linear(256, 256) ReLU Linear(256, d), per frame."""
    h = F.relu(F.linear(y, sd[f"{prefix}.0.weight"], sd[f"{prefix}.0.bias"]))
    h = F.relu(F.linear(h, sd[f"{prefix}.3.weight"], sd[f"{prefix}.3.bias"]))
    return F.linear(h, sd[f"{prefix}.6.weight"], sd[f"{prefix}.6.bias"])


def cross_mha(sd, prefix: str, x, mem, nhead: int):
    """MultiheadAttention.forward(x, mem, mem) (valle/modules/activation.py:199-431 -> F.multi_head_attention_forward): queries
    from the first d rows of the packed in-proj, keys / values of ``mem`` from the other 2 d, no mask.  x (T, d), mem (S, d)."""
    T, d = x.shape
    S = mem.shape[0]
    dh = d // nhead
    w, b = sd[f"{prefix}.in_proj_weight"], sd[f"{prefix}.in_proj_bias"]
    q = F.linear(x, w[:d], b[:d]).view(T, nhead, dh).transpose(0, 1)
    k = F.linear(mem, w[d: 2 * d], b[d: 2 * d]).view(S, nhead, dh).transpose(0, 1)
    v = F.linear(mem, w[2 * d:], b[2 * d:]).view(S, nhead, dh).transpose(0, 1)
    p = torch.softmax(torch.matmul(q, k.transpose(1, 2)) / math.sqrt(dh), dim=-1)
    o = torch.matmul(p, v).transpose(0, 1).reshape(T, d)
    return F.linear(o, sd[f"{prefix}.out_proj.weight"], sd[f"{prefix}.out_proj.bias"])


def decoder_layer(sd, prefix: str, x, mem, nhead: int, tgt_mask, stage_emb, norm_first: bool):
    """TransformerDecoderLayer.forward (valle/modules/transformer.py:520-575): self-attention, cross-attention over the
    memory, FFN; pre-norm (:544-556) or post-norm (:557-571)."""
    n = lambda i, t: norm_site(sd, f"{prefix}.norm{i}", t, stage_emb)  # noqa: E731
    ff = lambda t: F.linear(F.relu(F.linear(t, sd[f"{prefix}.linear1.weight"], sd[f"{prefix}.linear1.bias"])),  # noqa: E731
                            sd[f"{prefix}.linear2.weight"], sd[f"{prefix}.linear2.bias"])
    if norm_first:
        x = x + mha(sd, f"{prefix}.self_attn", n(1, x), nhead, tgt_mask)
        x = x + cross_mha(sd, f"{prefix}.multihead_attn", n(2, x), mem, nhead)
        return x + ff(n(3, x))
    x = n(1, x + mha(sd, f"{prefix}.self_attn", x, nhead, tgt_mask))
    x = n(2, x + cross_mha(sd, f"{prefix}.multihead_attn", x, mem, nhead))
    return n(3, x + ff(x))


def decoder(sd, prefix: str, cfg: OracleConfig, tgt, mem, tgt_mask=None, stage_emb=None):
    """nn.TransformerDecoder.forward of torch 1.13 (the reference's pin, README.md:31): each layer on the previous output and the
    memory, then the final norm if there is one (valle.py:141-152, 232-246).  (torch >= 2's container rejects the reference's
    tuple inputs, SURVEY.md 8c: the goldens come from the reference's own layers under this loop, oracle/make_golden.py.)"""
    nar = prefix.startswith("nar_")
    x = tgt
    for l in range(cfg.nar_num_layers if nar else cfg.num_layers):
        x = decoder_layer(sd, f"{prefix}.layers.{l}", x, mem, cfg.nar_nhead if nar else cfg.nhead, tgt_mask, stage_emb, cfg.norm_first)
    return norm_site(sd, f"{prefix}.norm", x, stage_emb) if cfg.norm_first else x


def vallf_ar_decode(sd, cfg: OracleConfig, x_ids, x_lens, prompts, top_k=-100, temperature=1.0, max_new=None, trace=None, generator=None):
    """VALLF.inference's AR loop, valle.py:593-652: the text is the decoder's MEMORY, the audio stream its causal target."""
    assert x_ids.ndim == 2 and x_lens.ndim == 1 and prompts.ndim == 3 and prompts.shape[0] == 1  # :591-594
    assert torch.all(x_lens > 0)
    S = int(x_lens.max())
    x = token_embedding(sd, "ar_text_embedding", x_ids[0])  # :599
    if cfg.add_prenet:
        x = text_prenet(sd, "ar_text_prenet", x)
    x = sine_position(x, sd["ar_text_position.alpha"])  # :601
    P = prompts.shape[1]
    y = prompts[0, :, 0]  # :610
    if cfg.prepend_bos:
        y = F.pad(y, (1, 0), value=NUM_AUDIO_TOKENS + 1)  # :611-612
    bos = int(cfg.prepend_bos)
    while True:
        e = token_embedding(sd, "ar_audio_embedding", y)  # :615-617
        if cfg.add_prenet:
            e = audio_prenet(sd, "ar_audio_prenet", e)
        y_pos = sine_position(e, sd["ar_audio_position.alpha"])
        T = y.shape[0]
        tgt_mask = torch.triu(torch.ones(T, T, dtype=torch.bool), diagonal=1)  # :619-624
        y_dec = decoder(sd, "ar_decoder", cfg, y_pos, x, tgt_mask=tgt_mask)  # :626-632
        logits = F.linear(y_dec[-1:], sd["ar_predict_layer.weight"])  # :633
        if trace is not None:
            trace.setdefault("ar_logits", []).append(logits[0].clone())
        samples = topk_sampling(logits.clone(), top_k, temperature, generator)  # :634-636
        n_gen = T - bos - P
        stop = (int(torch.argmax(logits, dim=-1)[0]) == NUM_AUDIO_TOKENS or int(samples[0, 0]) == NUM_AUDIO_TOKENS
                or (T - P) > S * 16)  # :638-642
        if max_new is not None and n_gen >= max_new:
            stop = True
        if stop:
            if P == T:
                raise SyntaxError("well trained model shouldn't reach here.")  # :643-646
            break
        y = torch.cat([y, samples[0]], 0)  # :651
    return y[None]


def vallf_nar_decode(sd, cfg: OracleConfig, text_ids, y0, prompts, prefix_len, enroll_x_lens=None, trace=None):
    """VALLF.inference's NAR stages, valle.py:653-710: like VALL-E's, but the decoder runs over the audio stream only and
    attends to the text through cross-attention."""
    Q = cfg.num_quantizers
    codes = [y0[prefix_len:]]  # :653
    if Q == 1:
        return torch.stack(codes, dim=-1)[None]
    y_emb = token_embedding(sd, "nar_audio_embeddings.0", y0).clone()  # :658-660
    text = text_ids
    if cfg.prefix_mode in (2, 4):  # :661-671
        enrolled_len = int(enroll_x_lens.max())
        text = torch.cat([text[:1], text[enrolled_len - 1:]], 0)
    x = token_embedding(sd, "nar_text_embedding", text)  # :673-675
    if cfg.add_prenet:
        x = text_prenet(sd, "nar_text_prenet", x)
    x = sine_position(x, sd["nar_text_position.alpha"])
    if cfg.prefix_mode != 0:
        for j in range(1, Q):  # :677-681
            y_emb[:prefix_len] += token_embedding(sd, f"nar_audio_embeddings.{j}", prompts[:, j])
    for i in range(Q - 1):  # :683-705
        y_pos = audio_prenet(sd, "nar_audio_prenet", y_emb) if cfg.add_prenet else y_emb
        y_pos = sine_position(y_pos, sd["nar_audio_position.alpha"])
        stage = sd[f"nar_stage_embeddings.{i}.word_embeddings.weight"]
        y_dec = decoder(sd, "nar_decoder", cfg, y_pos, x, tgt_mask=None, stage_emb=stage)  # :691-697
        logits = F.linear(y_dec[prefix_len:], sd[f"nar_predict_layers.{i}.weight"])  # :698
        if trace is not None:
            trace.setdefault("nar_logits", []).append(logits.clone())
        samples = torch.argmax(logits, dim=-1)  # :699
        codes.append(samples)
        if i < Q - 2:  # :701 (``if i < 6``)
            if cfg.prefix_mode == 0:
                y_emb[:prefix_len] += token_embedding(sd, f"nar_audio_embeddings.{i + 1}", prompts[:, i + 1])
            y_emb[prefix_len:] += token_embedding(sd, f"nar_audio_embeddings.{i + 1}", samples)
    assert len(codes) == Q
    return torch.stack(codes, dim=-1)[None]


def prefix_lm_mask(S: int, T: int) -> torch.Tensor:
    """valle.py:1018-1033: [[0_{SxS} | 1_{SxT}], [0_{TxS} | triu(1)_{TxT}]], True = blocked."""
    top = F.pad(torch.zeros(S, S, dtype=torch.bool), (0, T), value=True)
    bot = F.pad(torch.triu(torch.ones(T, T, dtype=torch.bool), diagonal=1), (S, 0), value=False)
    return torch.cat([top, bot], 0)


def top_k_filtering(logits: torch.Tensor, top_k: int) -> torch.Tensor:
    """top_k_top_p_filtering with top_p = 1.0 (valle.py:1242-1284; callers fix top_p, :1041)."""
    if top_k > 0:
        top_k = min(max(top_k, 1), logits.size(-1))
        kth = torch.topk(logits, top_k)[0][..., -1, None]
        logits = logits.masked_fill(logits < kth, float("-inf"))
    return logits


def topk_sampling(logits: torch.Tensor, top_k: int, temperature: float, generator=None) -> torch.Tensor:
    """valle.py:1287-1302.  top_k == 1 is the reference's 'greedy' (SURVEY.md fact 4)."""
    if temperature != 1.0:
        logits = logits / temperature
    logits = top_k_filtering(logits, top_k)
    return torch.multinomial(F.softmax(logits, dim=-1), num_samples=1, generator=generator)


# --------------------------------------------------------------------------------------
# the path
# --------------------------------------------------------------------------------------
def ar_decode(
    sd, cfg: OracleConfig, x_ids, x_lens, prompts, top_k=-100, temperature=1.0,
    kv_cache=False, force_tokens: Optional[torch.Tensor] = None, max_new: Optional[int] = None,
    trace: Optional[Dict[str, List]] = None, generator=None, act_fp8: bool = False,
):
    """The AR ``while True`` loop, valle.py:993-1059.  Returns y (1, [bos+]P+G) int64.

    ``force_tokens`` (G,) teacher-forces the token history (the sampled token is still traced);
    ``max_new`` stops after that many tokens (emulates a trained model's EOS from outside,
    SURVEY.md 8c G1).  ``trace['ar_logits']`` collects the (1025,) fp32 logits of every step."""
    assert x_ids.ndim == 2 and x_lens.ndim == 1 and prompts.ndim == 3 and prompts.shape[0] == 1  # :986-989
    assert torch.all(x_lens > 0)  # :991
    S = int(x_lens.max())
    x = token_embedding(sd, "ar_text_embedding", x_ids[0])  # :994
    if cfg.add_prenet:
        x = text_prenet(sd, "ar_text_prenet", x)  # :996 (Identity without add_prenet, :124-126)
    x = sine_position(x, sd["ar_text_position.alpha"])  # :997

    def audio_in(ids, start=0):  # :1013-1015
        e = token_embedding(sd, "ar_audio_embedding", ids)
        if cfg.add_prenet:
            e = audio_prenet(sd, "ar_audio_prenet", e)
        return sine_position(e, sd["ar_audio_position.alpha"], start=start)

    P = prompts.shape[1]
    y = prompts[0, :, 0]  # :1005
    if cfg.prepend_bos:
        y = F.pad(y, (1, 0), value=NUM_AUDIO_TOKENS + 1)  # :1006-1007
    bos = int(cfg.prepend_bos)
    kv_states = [dict(k=None, v=None) for _ in range(cfg.num_layers)] if kv_cache else None
    n_cached_audio = 0
    while True:
        if not kv_cache:
            y_pos = audio_in(y)  # :1013-1015
            xy_pos = torch.cat([x, y_pos], 0)  # :1016
            mask = prefix_lm_mask(S, y.shape[0]).to(x.device)  # :1018-1033
            xy_dec = encoder(sd, "ar_decoder", cfg, xy_pos, attn_mask=mask)  # :1035-1038
        else:
            new = y[n_cached_audio:]
            y_pos = audio_in(new, n_cached_audio)
            if n_cached_audio == 0:
                inp = torch.cat([x, y_pos], 0)
                mask = prefix_lm_mask(S, y.shape[0]).to(x.device)
            else:
                inp, mask = y_pos, None  # the new row sees every cached key (causal row = last)
            # engine mode FP8: only the packed prefill pass quantises its activations; the single-row steps do not
            xy_dec = encoder(sd, "ar_decoder", cfg, inp, attn_mask=mask, kv_states=kv_states, act_fp8=act_fp8 and mask is not None)
            n_cached_audio = y.shape[0]
        logits = F.linear(xy_dec[-1:], sd["ar_predict_layer.weight"])  # :1039  (1, 1025)
        if trace is not None:
            trace.setdefault("ar_logits", []).append(logits[0].clone())
        # topk_sampling filters in place when temperature == 1 (valle.py:1260,1296-1299); the
        # argmax of filtered and unfiltered logits coincide, so a copy is equivalent.
        samples = topk_sampling(logits.clone(), top_k, temperature, generator)  # :1040-1042
        n_gen = y.shape[0] - bos - P
        stop = (
            int(torch.argmax(logits, dim=-1)[0]) == NUM_AUDIO_TOKENS
            or int(samples[0, 0]) == NUM_AUDIO_TOKENS
            or (y.shape[0] - P) > S * 16
        )  # :1044-1048
        if max_new is not None and n_gen >= max_new:
            stop = True
        if force_tokens is not None:  # instrumentation: the forced history alone decides the length
            stop = n_gen >= force_tokens.shape[0]
        if stop:
            if P == y.shape[0]:
                raise SyntaxError("well trained model shouldn't reach here.")  # :1049-1052
            break
        nxt = samples[0] if force_tokens is None else force_tokens[n_gen : n_gen + 1].to(torch.int64)
        y = torch.cat([y, nxt], 0)  # :1057
    return y[None]


def nar_decode(sd, cfg: OracleConfig, text_ids, y0, prompts, prefix_len, enroll_x_lens=None, trace=None, act_fp8: bool = False):
    """The seven NAR stages, valle.py:1062-1137 (also the body of continual(), :1176-1238).

    text_ids (S,) ; y0 (P+G,) first-codebook stream without BOS ; prompts (P,Q)."""
    Q = cfg.num_quantizers
    codes = [y0[prefix_len:]]  # :1059
    if Q == 1:
        return torch.stack(codes, dim=-1)[None]
    y_emb = token_embedding(sd, "nar_audio_embeddings.0", y0).clone()  # :1064-1066
    text = text_ids
    if cfg.prefix_mode in (2, 4):  # :1068-1079 -- drop the enrolled phonemes
        enrolled_len = int(enroll_x_lens.max())
        text = torch.cat([text[:1], text[enrolled_len - 1 :]], 0)
    S = text.shape[0]
    x = token_embedding(sd, "nar_text_embedding", text)  # :1081
    if cfg.add_prenet:
        x = text_prenet(sd, "nar_text_prenet", x)  # :1082
    x = sine_position(x, sd["nar_text_position.alpha"])  # :1083
    if cfg.prefix_mode != 0:
        for j in range(1, Q):  # :1110-1113
            y_emb[:prefix_len] += token_embedding(sd, f"nar_audio_embeddings.{j}", prompts[:, j])
    for i in range(Q - 1):  # :1085 / :1115
        y_pos = audio_prenet(sd, "nar_audio_prenet", y_emb) if cfg.add_prenet else y_emb  # :1121
        y_pos = sine_position(y_pos, sd["nar_audio_position.alpha"])  # :1122
        xy_pos = torch.cat([x, y_pos], 0)  # :1123
        stage = sd[f"nar_stage_embeddings.{i}.word_embeddings.weight"]  # (1, d)  :1126
        xy_dec = encoder(sd, "nar_decoder", cfg, xy_pos, attn_mask=None, stage_emb=stage, act_fp8=act_fp8)  # :1125-1127
        logits = F.linear(xy_dec[S + prefix_len :], sd[f"nar_predict_layers.{i}.weight"])  # :1128
        if trace is not None:
            trace.setdefault("nar_logits", []).append(logits.clone())
        samples = torch.argmax(logits, dim=-1)  # :1130
        codes.append(samples)
        if i < Q - 2:  # :1133 / :1103
            if cfg.prefix_mode == 0:
                y_emb[:prefix_len] += token_embedding(sd, f"nar_audio_embeddings.{i + 1}", prompts[:, i + 1])  # :1104-1107
            y_emb[prefix_len:] += token_embedding(sd, f"nar_audio_embeddings.{i + 1}", samples)  # :1108 / :1134
    assert len(codes) == Q
    return torch.stack(codes, dim=-1)[None]  # :1136-1137


@torch.no_grad()
def inference(
    sd, cfg: OracleConfig, x, x_lens, y, enroll_x_lens=None, top_k=-100, temperature=1.0,
    kv_cache=False, force_tokens=None, max_new=None, trace=None, generator=None, quiet=True, act_fp8=False,
):
    """VALLE.inference, valle/models/valle.py:961-1137.  x (1,S) int64, x_lens (1,) int32,
    y (1,P,Q) int64 -> (1,G,Q) int64."""
    cfg.check_supported()
    if cfg.model == "vallf":  # VALLF.inference, valle.py:566-710 (no incremental mode: the literal loop only)
        assert not act_fp8 and force_tokens is None
        yy = vallf_ar_decode(sd, cfg, x, x_lens, y, top_k, temperature, max_new, trace, generator)
        return vallf_nar_decode(sd, cfg, x[0], yy[0, int(cfg.prepend_bos):], y[0], y.shape[1], enroll_x_lens, trace)
    assert not act_fp8 or kv_cache, "act_fp8 models the engine's packed prefill + single-row steps: needs kv_cache=True"
    yy = ar_decode(sd, cfg, x, x_lens, y, top_k, temperature, kv_cache, force_tokens, max_new, trace, generator, act_fp8)
    P = y.shape[1]
    if not quiet:
        print(f"VALL-E EOS [{P} -> {yy.shape[1]}]")  # :1054
    y0 = yy[0, int(cfg.prepend_bos) :]
    return nar_decode(sd, cfg, x[0], y0, y[0], P, enroll_x_lens, trace, act_fp8)


@torch.no_grad()
def continual(sd, cfg: OracleConfig, x, x_lens, y, trace=None):
    """VALLE.continual, valle.py:1139-1238: NAR only; first codebook taken from y[...,0];
    prefix = min(T/2, 225) (:1173)."""
    cfg.check_supported()
    assert x.ndim == 2 and x_lens.ndim == 1 and y.ndim == 3 and y.shape[0] == 1
    assert torch.all(x_lens > 0)
    assert cfg.num_quantizers == 8  # :1160
    prefix_len = min(int(y.shape[1] * 0.5), 3 * 75)
    prompts = y[0, :prefix_len]
    c = OracleConfig(**{**cfg.__dict__})
    if c.prefix_mode in (2, 4):
        c.prefix_mode = 1  # continual() never slices the text (no :1068-1079 equivalent)
    return nar_decode(sd, c, x[0], y[0, :, 0], prompts, prefix_len, None, trace)


# --------------------------------------------------------------------------------------
# algorithmic byte / flop model of SURVEY.md 8(d) (used by bench.py for `roofline.achieved`)
# --------------------------------------------------------------------------------------
def expected_gen_len(S: int, prepend_bos: bool = False) -> int:
    """Length cap of the stop rule when EOS never fires (valle.py:1047): G = 16*S + 1 - bos."""
    return 16 * S + 1 - int(prepend_bos)


# --------------------------------------------------------------------------------------
# FP8W weight format (engine mode VLE_DTYPE_FP8W) -- torch restatement used as the checker
# --------------------------------------------------------------------------------------
FP8W_MAX = 448.0  # largest finite e4m3fn value


def fp8w_quantize(w: torch.Tensor):
    """Per row of w (N, K) fp32: scale = smallest power of two with max|w_row| / scale <= 448,
    q = w / scale rounded to torch.float8_e4m3fn (round-to-nearest-even), W' = q * scale.
    Returns (q as uint8 bit patterns, scale fp32 (N,), W' fp32 (N, K))."""
    w = w.detach().to(torch.float32)
    amax = w.abs().amax(dim=1)
    m, ex = torch.frexp(amax / FP8W_MAX)  # amax / 448 = m * 2^ex, m in [0.5, 1)
    ex = torch.where(m == 0.5, ex - 1, ex)
    scale = torch.where(amax > 0, torch.ldexp(torch.ones_like(amax), ex), torch.ones_like(amax))
    q = (w / scale[:, None]).to(torch.float8_e4m3fn)
    return q.view(torch.uint8), scale, q.to(torch.float32) * scale[:, None]


def fp8w_state_dict(sd):
    """The state dict the FP8W engine computes with: every nn.Linear weight of the two decoders and of the
    predict layers replaced by W'; embeddings (also the ones a predict layer is tied to), LayerNorm affines,
    biases and the AdaLN projections unchanged (the engine folds those in fp32)."""
    out = {}
    for k, v in sd.items():
        lin = (k.endswith("self_attn.in_proj_weight") or k.endswith("self_attn.out_proj.weight") or k.endswith("linear1.weight")
               or k.endswith("linear2.weight") or k == "ar_predict_layer.weight" or (k.startswith("nar_predict_layers.") and k.endswith(".weight")))
        out[k] = fp8w_quantize(v)[2] if lin else v.clone()
    return out


# --------------------------------------------------------------------------------------
# teacher-forced forward: VALLE.forward, valle/models/valle.py:762-959, in eval mode
# --------------------------------------------------------------------------------------
def _topk_accuracy(logits: torch.Tensor, targets: torch.Tensor, k: int = 10, ignore_index: int = NUM_AUDIO_TOKENS) -> torch.Tensor:
    """torchmetrics MulticlassAccuracy(top_k=10, average="micro", ignore_index=1024) as constructed at valle.py:157-163,
    273-279 (third-party, not installed: restated from its published definition -- this metric is parity-UNPINNED).
    logits (rows, C), targets (rows,)."""
    hits, keep = _topk_counts(logits, targets, k, ignore_index)
    return hits / keep.clamp_min(1)


def _topk_counts(logits: torch.Tensor, targets: torch.Tensor, k: int = 10, ignore_index: int = NUM_AUDIO_TOKENS):
    top = logits.topk(min(k, logits.shape[1]), dim=1).indices
    hit = (top == targets[:, None]).any(dim=1)
    keep = targets != ignore_index
    return (hit & keep).sum().float(), keep.sum().float()


@torch.no_grad()
def forward(sd, cfg: OracleConfig, x, x_lens, y, y_lens, reduction: str = "sum", train_stage: int = 0,
            nar_stage: Optional[int] = None, prefix_len: Optional[int] = None, trace=None, prompt_starts=None, y_prompts=None):
    """VALLE.forward (valle.py:762-959) for UNPADDED batches (every x_lens == x.shape[1], every y_lens == y.shape[1]), every
    prefix_mode, dropout off (eval).  The random draws of the reference are explicit arguments: ``nar_stage`` (self.rng.choices,
    :891-895); prefix_mode 1's ``prefix_len`` (torch.randint, :348-350); prefix_mode 2's per-utterance segment starts
    ``prompt_starts`` (self.rng.randint, :368-369).  prefix_mode 4's prompts (the PromptedFeatures' first half, :792-798) are
    ``y_prompts`` (N, P, Q).  Returns (total_loss, metrics) like the last two elements of the reference's tuple."""
    cfg.check_supported()
    assert x.ndim == 2 and x_lens.ndim == 1 and y.ndim == 3 and y_lens.ndim == 1  # :789-790, 801-802
    assert reduction == "sum"
    assert cfg.prefix_mode in (0, 1, 2, 4)
    N, S = x.shape
    T = y.shape[1]
    if not (all(int(v) == S for v in x_lens) and all(int(v) == T for v in y_lens)):
        return forward_padded(sd, cfg, x, x_lens, y, y_lens, train_stage, nar_stage, prefix_len, prompt_starts, y_prompts)
    codes = y.to(torch.int64)
    bos = int(cfg.prepend_bos)
    total = torch.zeros(())
    metrics = {}
    if train_stage in (0, 1):
        ar_loss = torch.zeros(())
        hits, kept = torch.zeros(()), torch.zeros(())
        for b in range(N):
            yb = codes[b, :, 0]
            targets = torch.cat([yb[1:], torch.tensor([NUM_AUDIO_TOKENS])]) if not bos else torch.cat([yb, torch.tensor([NUM_AUDIO_TOKENS])])  # pad_y_eos :322-333
            inputs = yb if not bos else F.pad(yb, (1, 0), value=NUM_AUDIO_TOKENS + 1)
            xe, ye = token_embedding(sd, "ar_text_embedding", x[b]), token_embedding(sd, "ar_audio_embedding", inputs)
            if cfg.add_prenet:
                xe, ye = text_prenet(sd, "ar_text_prenet", xe), audio_prenet(sd, "ar_audio_prenet", ye)  # :828, :862
            xe = sine_position(xe, sd["ar_text_position.alpha"])  # :827-829
            ye = sine_position(ye, sd["ar_audio_position.alpha"])  # :861-863
            if cfg.model == "vallf":  # VALLF.forward, valle.py:474-489: the text is the memory, the audio a causal target
                Ta = inputs.shape[0]
                dec = decoder(sd, "ar_decoder", cfg, ye, xe, tgt_mask=torch.triu(torch.ones(Ta, Ta, dtype=torch.bool), diagonal=1))
            else:
                mask = prefix_lm_mask(S, inputs.shape[0])  # :833-859 without padding
                dec = encoder(sd, "ar_decoder", cfg, torch.cat([xe, ye], 0), attn_mask=mask)[S:]  # :867-872
            logits = F.linear(dec, sd["ar_predict_layer.weight"])  # :873
            if trace is not None:
                trace.setdefault("ar_logits", []).append(logits.clone())
            ar_loss = ar_loss + F.cross_entropy(logits, targets, reduction="sum")  # :875
            h, kp = _topk_counts(logits, targets)
            hits, kept = hits + h, kept + kp
        total = total + ar_loss
        metrics["ArTop10Accuracy"] = float(hits / kept.clamp_min(1)) * float(N * T)  # :877-879 (micro accuracy x y_lens.sum())
    if cfg.num_quantizers == 1:
        return total, metrics
    if train_stage in (0, 2):
        assert nar_stage is not None and 1 <= nar_stage < cfg.num_quantizers
        P = 0
        if cfg.prefix_mode == 1:
            assert prefix_len is not None
            P = int(prefix_len)
        elif cfg.prefix_mode == 2:
            P = min(225, int(0.25 * T))  # :364
            assert prompt_starts is not None and len(prompt_starts) == N
        elif cfg.prefix_mode == 4:
            assert y_prompts is not None and y_prompts.shape[0] == N
            P = int(y_prompts.shape[1])  # :377
        nar_loss = torch.zeros(())
        hits, kept = torch.zeros(()), torch.zeros(())
        for b in range(N):
            if cfg.prefix_mode in (2, 4):
                # _prepare_prompts :362-389: the prompt is a separate segment IN FRONT of the whole utterance; prefix_mode 2 cuts
                # it out of the utterance itself and blanks that stretch of the TARGET codebook (ignore_index), :370-373
                cb = codes[b].clone()
                if cfg.prefix_mode == 2:
                    st = int(prompt_starts[b])
                    assert 0 <= st <= T - P
                    pr = cb[st: st + P].clone()
                    cb[st: st + P, nar_stage] = NUM_AUDIO_TOKENS
                else:
                    pr = y_prompts[b].to(torch.int64)
                xe = token_embedding(sd, "nar_text_embedding", x[b])
                if cfg.add_prenet:
                    xe = text_prenet(sd, "nar_text_prenet", xe)
                xe = sine_position(xe, sd["nar_text_position.alpha"])
                y_pr = token_embedding(sd, "nar_audio_embeddings.0", pr[:, 0]).clone()
                y_emb = token_embedding(sd, "nar_audio_embeddings.0", cb[:, 0]).clone()
                for j in range(1, cfg.num_quantizers):
                    y_pr += token_embedding(sd, f"nar_audio_embeddings.{j}", pr[:, j])
                    if j < nar_stage:
                        y_emb += token_embedding(sd, f"nar_audio_embeddings.{j}", cb[:, j])
                y_emb = torch.cat([y_pr, y_emb], 0)  # :389
                targets = cb[:, nar_stage]           # :906 (the whole utterance; the blanked stretch is ignored by the loss)
                ye = sine_position(audio_prenet(sd, "nar_audio_prenet", y_emb) if cfg.add_prenet else y_emb, sd["nar_audio_position.alpha"])
                stage = sd[f"nar_stage_embeddings.{nar_stage - 1}.word_embeddings.weight"]
                if cfg.model == "vallf":
                    dec = decoder(sd, "nar_decoder", cfg, ye, xe, tgt_mask=None, stage_emb=stage)[P:]  # :531-533
                else:
                    dec = encoder(sd, "nar_decoder", cfg, torch.cat([xe, ye], 0), attn_mask=None, stage_emb=stage)[S + P:]  # :927
                logits = F.linear(dec, sd[f"nar_predict_layers.{nar_stage - 1}.weight"])
                if trace is not None:
                    trace.setdefault("nar_logits", []).append(logits.clone())
                nar_loss = nar_loss + F.cross_entropy(logits, targets, ignore_index=NUM_AUDIO_TOKENS, reduction="sum")
                h, kp = _topk_counts(logits, targets)
                hits, kept = hits + h, kept + kp
                continue
            y0 = codes[b, :, 0]
            xe = token_embedding(sd, "nar_text_embedding", x[b])
            if cfg.add_prenet:
                xe = text_prenet(sd, "nar_text_prenet", xe)  # :898
            xe = sine_position(xe, sd["nar_text_position.alpha"])  # :897-899
            y_emb = token_embedding(sd, "nar_audio_embeddings.0", y0).clone()  # _prepare_prompts :335-393
            if cfg.prefix_mode == 0:
                for j in range(1, nar_stage):
                    y_emb += token_embedding(sd, f"nar_audio_embeddings.{j}", codes[b, :, j])
            else:
                for j in range(1, cfg.num_quantizers):
                    y_emb[:P] += token_embedding(sd, f"nar_audio_embeddings.{j}", codes[b, :P, j])
                    if j < nar_stage:
                        y_emb[P:] += token_embedding(sd, f"nar_audio_embeddings.{j}", codes[b, P:, j])
            targets = codes[b, P:, nar_stage]  # :906, :916-917
            ye = sine_position(audio_prenet(sd, "nar_audio_prenet", y_emb) if cfg.add_prenet else y_emb, sd["nar_audio_position.alpha"])  # :919-920
            stage = sd[f"nar_stage_embeddings.{nar_stage - 1}.word_embeddings.weight"]
            if cfg.model == "vallf":  # valle.py:537-544: cross-attention to the text
                dec = decoder(sd, "nar_decoder", cfg, ye, xe, tgt_mask=None, stage_emb=stage)[P:]
            else:
                dec = encoder(sd, "nar_decoder", cfg, torch.cat([xe, ye], 0), attn_mask=None, stage_emb=stage)[S + P:]  # :922-926
            logits = F.linear(dec, sd[f"nar_predict_layers.{nar_stage - 1}.weight"])  # :927-932
            if trace is not None:
                trace.setdefault("nar_logits", []).append(logits.clone())
            nar_loss = nar_loss + F.cross_entropy(logits, targets, reduction="sum")  # :936-942 (no padded targets here)
            h, kp = _topk_counts(logits, targets)  # the reference pads a 1025th class with the global minimum: never in the top 10
            hits, kept = hits + h, kept + kp
        total_length = float(N * T)
        if cfg.prefix_mode == 4:
            P = 0  # :929-930 "reset for Top10Accuracy metric" -- which also resets the loss's length correction
        total = total + nar_loss * (total_length / (total_length - P * N))  # :943
        metrics["NarTop10Accuracy"] = float(hits / kept.clamp_min(1)) * total_length  # :945-956
    if train_stage == 0:
        total = total / 2.0  # :958-959
    return total, metrics


@torch.no_grad()
def forward_padded(sd, cfg: OracleConfig, x, x_lens, y, y_lens, train_stage: int = 0, nar_stage: Optional[int] = None,
                   prefix_len: Optional[int] = None, prompt_starts=None, y_prompts=None):
    """VALLE.forward (valle.py:762-959) on a PADDED batch (x_lens.max() == x.shape[1], y_lens.max() == y.shape[1], the collater's
    shapes), literally: padded frames' codes are zeroed (:811), padded first-codebook inputs / targets are EOS (pad_y_eos,
    :322-333), padded KEYS are masked in every attention (:846-856, :908-926) -- and padded QUERY rows are computed like any
    other, because the AR loss (:875, no ignore_index) sums over them with target EOS.  VALL-E only (model "valle")."""
    assert cfg.model != "vallf", "padded forward: VALL-E only"
    N, S = x.shape
    T = y.shape[1]
    x_lens, y_lens = x_lens.to(torch.int64), y_lens.to(torch.int64)
    assert int(x_lens.max()) == S and int(y_lens.max()) == T, "the collater pads to the longest utterance (make_pad_mask sizes to it)"
    x_mask = torch.arange(S)[None] >= x_lens[:, None]  # make_pad_mask, :805-806
    y_mask = torch.arange(T)[None] >= y_lens[:, None]
    codes = y.to(torch.int64) * (~y_mask)[..., None].to(torch.int64)  # :811
    bos = int(cfg.prepend_bos)
    t_all = F.pad(codes[..., 0], (0, 1), value=0) + NUM_AUDIO_TOKENS * F.pad(y_mask.to(torch.int64), (0, 1), value=1)  # pad_y_eos
    if bos:
        inputs, targets = F.pad(t_all[:, :-1], (1, 0), value=NUM_AUDIO_TOKENS + 1), t_all
    else:
        inputs, targets = t_all[:, :-1], t_all[:, 1:]
    total = torch.zeros(())
    metrics = {}
    total_length = float(y_lens.sum())
    if train_stage in (0, 1):
        ar_y_mask = F.pad(y_mask, (1, 0), value=False) if bos else y_mask  # :821-826
        Ta = inputs.shape[1]
        ar_loss, hits, kept = torch.zeros(()), torch.zeros(()), torch.zeros(())
        for b in range(N):
            xe, ye = token_embedding(sd, "ar_text_embedding", x[b]), token_embedding(sd, "ar_audio_embedding", inputs[b])
            if cfg.add_prenet:
                xe, ye = text_prenet(sd, "ar_text_prenet", xe), audio_prenet(sd, "ar_audio_prenet", ye)
            xe = sine_position(xe, sd["ar_text_position.alpha"])
            ye = sine_position(ye, sd["ar_audio_position.alpha"])
            mask = prefix_lm_mask(S, Ta) | torch.cat([x_mask[b], ar_y_mask[b]])[None, :]  # :846-852
            dec = encoder(sd, "ar_decoder", cfg, torch.cat([xe, ye], 0), attn_mask=mask)[S:]
            logits = F.linear(dec, sd["ar_predict_layer.weight"])
            ar_loss = ar_loss + F.cross_entropy(logits, targets[b], reduction="sum")  # :875: padded rows included
            h, kp = _topk_counts(logits, targets[b])
            hits, kept = hits + h, kept + kp
        total = total + ar_loss
        metrics["ArTop10Accuracy"] = float(hits / kept.clamp_min(1)) * total_length
    if cfg.num_quantizers == 1:
        return total, metrics
    y_in = inputs[:, 1:] if bos else inputs  # :886-887
    if train_stage in (0, 2):
        assert nar_stage is not None and 1 <= nar_stage < cfg.num_quantizers
        ymin = int(y_lens.min())
        P = 0
        if cfg.prefix_mode == 1:
            assert prefix_len is not None
            P = int(prefix_len)
        elif cfg.prefix_mode == 2:
            P = min(225, int(0.25 * ymin))
            assert prompt_starts is not None and len(prompt_starts) == N
        elif cfg.prefix_mode == 4:
            assert y_prompts is not None
            P = int(y_prompts.shape[1])
        nar_loss, hits, kept = torch.zeros(()), torch.zeros(()), torch.zeros(())
        for b in range(N):
            cb = codes[b].clone()
            xe = token_embedding(sd, "nar_text_embedding", x[b])
            if cfg.add_prenet:
                xe = text_prenet(sd, "nar_text_prenet", xe)
            xe = sine_position(xe, sd["nar_text_position.alpha"])
            if cfg.prefix_mode in (2, 4):
                if cfg.prefix_mode == 2:
                    st = int(prompt_starts[b])
                    assert 0 <= st <= int(y_lens[b]) - P
                    pr = cb[st: st + P].clone()
                    cb[st: st + P, nar_stage] = NUM_AUDIO_TOKENS
                else:
                    pr = y_prompts[b].to(torch.int64)
                y_pr = token_embedding(sd, "nar_audio_embeddings.0", pr[:, 0]).clone()
                y_emb = token_embedding(sd, "nar_audio_embeddings.0", y_in[b]).clone()
                for j in range(1, cfg.num_quantizers):
                    y_pr += token_embedding(sd, f"nar_audio_embeddings.{j}", pr[:, j])
                    if j < nar_stage:
                        y_emb += token_embedding(sd, f"nar_audio_embeddings.{j}", cb[:, j])
                y_emb = torch.cat([y_pr, y_emb], 0)
                kpm = torch.cat([x_mask[b], F.pad(y_mask[b], (P, 0), value=False)])  # :908-915
                tgt = cb[:, nar_stage] + NUM_AUDIO_TOKENS * y_mask[b].to(torch.int64)  # :906
                lo = S + P
            else:
                y_emb = token_embedding(sd, "nar_audio_embeddings.0", y_in[b]).clone()
                if cfg.prefix_mode == 0:
                    for j in range(1, nar_stage):
                        y_emb += token_embedding(sd, f"nar_audio_embeddings.{j}", cb[:, j])
                else:
                    for j in range(1, cfg.num_quantizers):
                        y_emb[:P] += token_embedding(sd, f"nar_audio_embeddings.{j}", cb[:P, j])
                        if j < nar_stage:
                            y_emb[P:] += token_embedding(sd, f"nar_audio_embeddings.{j}", cb[P:, j])
                kpm = torch.cat([x_mask[b], y_mask[b]])
                tgt = (cb[:, nar_stage] + NUM_AUDIO_TOKENS * y_mask[b].to(torch.int64))[P:]  # :906, :916-917
                lo = S + P
            ye = sine_position(audio_prenet(sd, "nar_audio_prenet", y_emb) if cfg.add_prenet else y_emb, sd["nar_audio_position.alpha"])
            stage = sd[f"nar_stage_embeddings.{nar_stage - 1}.word_embeddings.weight"]
            Ltot = S + ye.shape[0]
            mask = kpm[None, :].expand(Ltot, Ltot)  # src_key_padding_mask only (:922-926)
            dec = encoder(sd, "nar_decoder", cfg, torch.cat([xe, ye], 0), attn_mask=mask, stage_emb=stage)[lo:]
            logits = F.linear(dec, sd[f"nar_predict_layers.{nar_stage - 1}.weight"])
            nar_loss = nar_loss + F.cross_entropy(logits, tgt, ignore_index=NUM_AUDIO_TOKENS, reduction="sum")
            h, kp = _topk_counts(logits, tgt)
            hits, kept = hits + h, kept + kp
        if cfg.prefix_mode == 4:
            P = 0
        total = total + nar_loss * (total_length / (total_length - P * N))  # :943
        metrics["NarTop10Accuracy"] = float(hits / kept.clamp_min(1)) * total_length
    if train_stage == 0:
        total = total / 2.0
    return total, metrics
