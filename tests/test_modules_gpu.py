"""GPU parity of the block API (valle_amd/modules.py) against golden vectors produced by the reference's
own modules (oracle/make_golden_modules.py -> tests/golden/modules), plus a cross-check of the two HIP
paths: the reference's decode loop written against the block modules must give the engine's tokens.

Tolerances: fp32 kernels |err| <= 2e-4 * max(1, max|ref|) (fp32 MFMA / reduction-order noise);
bf16 kernels |err| <= 5 % of the output's standard deviation (SURVEY.md 8c G2 calibration)."""
import os

import numpy as np
import pytest
import torch

import valle_amd
from valle_amd import modules as M
from oracle import valle_oracle as vo
from oracle.make_golden_modules import ENCODER_CASES, build_encoder, fill_module_, prefix_lm_mask, rand

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "modules")


def gold(name):
    return np.load(os.path.join(GOLD, name))


def close(got: torch.Tensor, want: np.ndarray, dtype: str, what: str = ""):
    want = torch.from_numpy(np.asarray(want))
    got = got.detach().float().cpu()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = (got - want).abs().max().item()
    tol = 2e-4 * max(1.0, want.abs().max().item()) if dtype == "fp32" else 0.05 * want.std().item()
    assert err <= tol, f"{what}: max|err| {err:.3e} > {tol:.3e} ({dtype})"


def test_token_and_positional_embedding():
    z = gold("embedding.npz")
    te = M.TokenEmbedding(64, 100).eval()
    fill_module_(te, 11)
    te = te.cuda()
    got = te(torch.from_numpy(z["ids"]).cuda())
    assert torch.equal(got.cpu(), torch.from_numpy(z["tok"]))  # a gather is exact
    for tag, kw in (("a", dict(alpha=True)), ("s", dict(scale=True, alpha=False))):
        pe = M.SinePositionalEmbedding(64, dropout=0.1, **kw).eval()
        fill_module_(pe, 12)
        pe = pe.cuda()
        got = pe(rand((2, 9, 64), 6).cuda())
        # same mul/mul/add order on the same kind of fp32 table; the table itself is built by the host's libm
        # (sin/cos/exp differ in the last ulp between the golden's host and this one): 2 ulp of |x| ~ 8
        assert (got.cpu() - torch.from_numpy(z[f"pos_{tag}"])).abs().max().item() <= 2e-6, tag


def test_layernorm_and_adaptive_layernorm():
    z = gold("norms.npz")
    ln = M.LayerNorm(64).eval()
    fill_module_(ln, 21)
    ln = ln.cuda()
    x = (rand((2, 5, 64), 7) * 3 + 0.5).cuda()
    emb = rand((1, 64), 8).cuda()
    close(ln(x), z["ln"], "fp32", "LayerNorm")
    y, e = ln((x, emb))
    assert e is emb
    close(y, z["ln_tuple"], "fp32", "LayerNorm tuple")
    with pytest.raises(AssertionError):
        ln(x, emb)  # transformer.py:69
    ada = M.AdaptiveLayerNorm(64, M.LayerNorm(64)).eval()
    fill_module_(ada, 22)
    ada = ada.cuda()
    close(ada(x, emb), z["ada"], "fp32", "AdaptiveLayerNorm")
    close(ada((x, emb))[0], z["ada_tuple"], "fp32", "AdaptiveLayerNorm tuple")


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_multihead_attention(dtype):
    z = gold("mha.npz")
    mha = M.MultiheadAttention(64, 4, dropout=0.1, batch_first=True).eval()
    fill_module_(mha, 31)
    mha = M.set_compute_dtype(mha.cuda(), dtype)
    x = rand((2, 11, 64), 9).cuda()
    o, w = mha(x, x, x, need_weights=False)
    assert w is None
    close(o, z["none"], dtype, "no mask")
    close(mha(x, x, x, need_weights=False, attn_mask=prefix_lm_mask(0, 11).cuda())[0], z["causal"], dtype, "causal")
    close(mha(x, x, x, need_weights=False, attn_mask=prefix_lm_mask(4, 11).cuda())[0], z["prefix4"], dtype, "prefix-LM")
    fm = torch.zeros(11, 11).masked_fill(prefix_lm_mask(4, 11), float("-inf")).cuda()
    close(mha(x, x, x, need_weights=False, attn_mask=fm)[0], z["prefix4_float"], dtype, "prefix-LM float mask")
    tf = M.MultiheadAttention(64, 4, dropout=0.0, batch_first=False).eval()
    fill_module_(tf, 31)
    tf = M.set_compute_dtype(tf.cuda(), dtype)
    xt = x.transpose(0, 1).contiguous()
    close(tf(xt, xt, xt, need_weights=False)[0], z["time_first"], dtype, "time-first")
    with pytest.raises(NotImplementedError):
        mha(x, x, x)  # need_weights=True
    with pytest.raises(NotImplementedError):
        mha(x, x.clone(), x, need_weights=False)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("name", sorted(ENCODER_CASES))
def test_transformer_encoder(name, dtype):
    d, nhead, layers, adaptive, B, T, S = ENCODER_CASES[name]
    z = gold(f"{name}.npz")
    enc = build_encoder(M, M, d, nhead, layers, adaptive)
    fill_module_(enc, 41)
    enc = M.set_compute_dtype(enc.cuda(), dtype)
    x = rand((B, T, d), 10).cuda()
    emb = rand((1, d), 11).cuda() if adaptive else None
    mask = prefix_lm_mask(S, T).cuda() if S is not None else None
    y, e = enc((x, emb), mask=mask)
    assert e is emb
    close(y, z["out"], dtype, "encoder output")
    states, (y2, _) = enc((x, emb), mask=mask, return_layer_states=True)
    assert torch.equal(y, y2) and len(states) == layers
    for i, s in enumerate(states):
        close(s, z[f"state{i}"], dtype, f"layer state {i}")
    if not adaptive:
        close(enc(x, mask=mask), z["out_plain"], dtype, "tensor in => tensor out")
    close(enc.layers[0]((x, emb), src_mask=mask)[0], z["layer0"], dtype, "layer 0")


def _module_level_inference(m: valle_amd.VALLE, x, y, max_new):
    """The reference's inference() loop (valle/models/valle.py:994-1134, prefix_mode 1, greedy, no KV cache)
    written against the block modules -- arg-max on logits is the only arithmetic done outside vle_op_*."""
    from valle_amd import ops

    def predict(h2d, weight):
        return ops.linear(h2d.contiguous(), weight.detach(), None, epilogue=ops.EPI_F32)

    text = m.ar_text_position(m.ar_text_embedding(x))
    S, P = x.shape[1], y.shape[1]
    ys = y[..., 0]
    while True:
        y_pos = m.ar_audio_position(m.ar_audio_embedding(ys))
        xy = torch.cat([text, y_pos], dim=1)
        T = xy.shape[1]
        h, _ = m.ar_decoder((xy, None), mask=prefix_lm_mask(S, T).to(xy.device))
        logits = predict(h[0, -1:], m.ar_predict_layer.weight)
        tok = int(logits.argmax(-1))
        if tok == 1024 or ys.shape[1] - P >= max_new:
            break
        ys = torch.cat([ys, torch.tensor([[tok]], device=ys.device)], dim=1)
    codes = [ys[:, P:]]
    G = ys.shape[1] - P
    xn = m.nar_text_position(m.nar_text_embedding(x))
    y_emb = m.nar_audio_embeddings[0](ys)
    for j in range(1, 8):  # prefix_mode != 0: prompt rows carry all 8 codebooks (valle.py:1110-1113)
        y_emb[:, :P] += m.nar_audio_embeddings[j](y[..., j])
    for i in range(7):
        y_pos = m.nar_audio_position(y_emb)
        xy = torch.cat([xn, y_pos], dim=1)
        h, _ = m.nar_decoder((xy, m.nar_stage_embeddings[i].weight))
        logits = predict(h[0, S + P:], m.nar_predict_layers[i].weight)
        samples = logits.argmax(-1)[None]
        codes.append(samples)
        if i < 6:
            y_emb[:, P:] += m.nar_audio_embeddings[i + 1](samples)  # valle.py:1134
    return torch.stack(codes, dim=-1), G


def test_block_modules_reproduce_the_engine_tokens():
    cfg = vo.OracleConfig(d_model=128, nhead=2, num_layers=2, prefix_mode=1)
    sd = vo.make_state_dict(cfg, 0)
    x, x_lens, y = vo.make_inputs(7, 12)
    m = valle_amd.VALLE(cfg.d_model, cfg.nhead, cfg.num_layers, prefix_mode=1, engine_dtype="fp32")
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    max_new = 24
    want = m.inference_batch(x.cuda(), x_lens, y.cuda(), [y.shape[1]], None, top_k=1, max_new=max_new)[0][None]
    got, G = _module_level_inference(m, x.cuda(), y.cuda(), max_new)
    assert G == want.shape[1]
    assert torch.equal(got.cpu(), want.cpu()), "block-module decode and engine decode disagree (fp32, greedy)"
    ref = vo.inference(sd, cfg, x, x_lens, y, None, top_k=1, kv_cache=True, max_new=max_new)
    assert torch.equal(got.cpu(), ref), "block-module decode differs from the oracle"


@pytest.mark.parametrize("mask_kind", ["none", "prefix_lm"])
def test_multihead_attention_key_padding_mask(mask_kind):
    """``key_padding_mask`` of the block MultiheadAttention (the teacher-forced forward's src_key_padding_mask, valle.py:846-856,
    :908-926): padded keys are invisible; VALID query rows equal the fp64 definition softmax((QK^T)/sqrt(dh) + masks) V with the
    reference's merged mask, for padding in the MIDDLE of a sequence (text pad | audio pad) too; padded AUDIO rows (the AR loss
    reads them) see exactly the valid keys of their sequence."""
    import math

    DEV = "cuda:0"
    torch.manual_seed(3)
    B, S, T, d, H = 3, 5, 9, 64, 4
    L = S + T
    mha = M.MultiheadAttention(d, H, batch_first=True).to(DEV).eval()
    M.set_compute_dtype(mha, "fp32")
    x = torch.randn(B, L, d, device=DEV)
    x_lens, y_lens = [5, 3, 4], [9, 6, 9]
    pad = torch.zeros(B, L, dtype=torch.bool)
    for b in range(B):
        pad[b, x_lens[b]:S] = True
        pad[b, S + y_lens[b]:] = True
    attn_mask = vo.prefix_lm_mask(S, T).to(DEV) if mask_kind == "prefix_lm" else None
    out, _ = mha(x, x, x, key_padding_mask=pad.to(DEV), need_weights=False, attn_mask=attn_mask)
    w, bias = mha.in_proj_weight.detach().double().cpu(), mha.in_proj_bias.detach().double().cpu()
    wo, bo = mha.out_proj.weight.detach().double().cpu(), mha.out_proj.bias.detach().double().cpu()
    dh = d // H
    for b in range(B):
        qkv = x[b].double().cpu() @ w.t() + bias
        q, k, v = (t.view(L, H, dh).transpose(0, 1) for t in (qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]))
        sc = q @ k.transpose(1, 2) / math.sqrt(dh)
        blocked = pad[b][None, :].expand(L, L).clone()
        if attn_mask is not None:
            blocked |= attn_mask.cpu()
        sc = sc.masked_fill(blocked[None], float("-inf"))
        ref = (torch.softmax(sc, -1) @ v).transpose(0, 1).reshape(L, d) @ wo.t() + bo
        rows = (~pad[b]).clone()
        rows[S + y_lens[b]:] = True  # ... and the padded audio rows
        err = (out[b].double().cpu()[rows] - ref[rows]).abs().max().item()
        assert err < 2e-4, (b, err)
    assert torch.isfinite(out).all()
