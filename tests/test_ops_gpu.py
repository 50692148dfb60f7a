"""GPU: each stand-alone HIP operator against its plain PyTorch fp32 definition
(the floating-point kernels' reference; tolerances stated per test)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import valle_amd  # noqa: E402
from valle_amd import ops  # noqa: E402

DEV = "cuda:0"


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


@pytest.mark.parametrize("d", [64, 192, 1024, 1536])
@pytest.mark.parametrize("rows", [1, 7, 300])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_layernorm(d, rows, dt):
    x = _rand(rows, d, seed=1, scale=3.0) + 0.5
    g = _rand(d, seed=2) * 0.2 + 1.0
    b = _rand(d, seed=3) * 0.1
    out = ops.layernorm(x, g, b, out_dtype=dt)
    ref = F.layer_norm(x.double(), (d,), g.double(), b.double(), 1e-5)
    if dt == torch.float32:
        assert (out.double() - ref).abs().max().item() < 2e-5
    else:
        assert torch.equal(out, ref.float().to(torch.bfloat16)) or (out.double() - ref).abs().max().item() < 0.04


GEMM_SHAPES = [(1, 64, 64), (37, 192, 64), (130, 256, 256), (272, 3072, 1024), (1025, 1024, 4096), (300, 1025, 1024), (64, 4096, 1024),
               (2, 3072, 1024), (17, 1025, 1024), (48, 1024, 4096), (64, 1024, 4096), (33, 768, 256), (9, 1536, 1536), (1041, 4096, 1024),
               (17408, 1024, 1024)]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("epi", [ops.EPI_STORE, ops.EPI_RELU, ops.EPI_RESID, ops.EPI_F32])
def test_linear_gemm(M, N, K, dt, epi):
    a = _rand(M, K, seed=4).to(dt)
    w = (_rand(N, K, seed=5) / math.sqrt(K)).to(dt)
    bias = _rand(N, seed=6) * 0.1
    ref = a.double() @ w.double().t() + bias.double()
    if epi == ops.EPI_RELU:
        ref = ref.clamp_min(0)
    if epi == ops.EPI_RESID:
        r0 = _rand(M, N, seed=7)
        out = ops.linear(a, w, bias, epi, resid=r0.clone())
        ref = ref + r0.double()
    else:
        out = ops.linear(a, w, bias, epi)
    err = (out.double() - ref).abs().max().item()
    # fp32 MFMA = exact fp32 FMA chain: error ~ K * eps * |a||w|; bf16 output rounding 2^-9 relative
    tol = 3e-5 * math.sqrt(K / 64) if dt == torch.float32 or epi in (ops.EPI_RESID, ops.EPI_F32) and False else None
    if dt == torch.float32:
        assert err < 1e-4, err
    elif epi in (ops.EPI_RESID, ops.EPI_F32):
        assert err < 2e-3, err  # bf16 inputs are exact in fp32 accumulate
    else:
        assert err < 0.03 * max(1.0, ref.abs().max().item()), err


SK_SHAPES = [(3072, 1024), (1024, 4096), (1025, 1024), (192, 64), (64, 256), (4096, 1024), (768, 192)]


@pytest.mark.parametrize("N,K", SK_SHAPES)
@pytest.mark.parametrize("M", [1, 2, 3, 5, 8])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_linear_skinny(N, K, M, dt):
    x = _rand(M, K, seed=8) * 2 + 0.3
    w = (_rand(N, K, seed=9) / math.sqrt(K)).to(dt)
    bias = _rand(N, seed=10) * 0.1
    g = _rand(K, seed=11) * 0.2 + 1.0
    b = _rand(K, seed=12) * 0.1
    wd = w.double()
    tol = 1e-4 if dt == torch.float32 else 2e-4
    # plain + store / relu / resid
    out = ops.linear_skinny(x, w, bias, 0)
    ref = x.double() @ wd.t() + bias.double()
    assert (out.double() - ref).abs().max().item() < tol
    out = ops.linear_skinny(x, w, bias, 1)
    assert (out.double() - ref.clamp_min(0)).abs().max().item() < tol
    r0 = _rand(M, N, seed=13)
    out = ops.linear_skinny(x, w, bias, 2, resid=r0.clone())
    assert (out.double() - (ref + r0.double())).abs().max().item() < tol
    # fused LayerNorm prologue, no bias (the logits head)
    out = ops.linear_skinny(x, w, None, 0, gamma=g, beta=b)
    xn = F.layer_norm(x.double(), (K,), g.double(), b.double(), 1e-5)
    assert (out.double() - xn @ wd.t()).abs().max().item() < tol * 3


def _ref_attention(qkv, lens, text_lens, nhead, causal):
    d = qkv.shape[1] // 3
    dh = d // nhead
    outs = []
    off = 0
    for L, S in zip(lens, text_lens):
        q, k, v = qkv[off : off + L].double().split(d, dim=1)
        q = q.view(L, nhead, dh).transpose(0, 1)
        k = k.view(L, nhead, dh).transpose(0, 1)
        v = v.view(L, nhead, dh).transpose(0, 1)
        s = q @ k.transpose(1, 2) / math.sqrt(dh)
        if causal:
            i = torch.arange(L, device=qkv.device)[:, None]
            j = torch.arange(L, device=qkv.device)[None, :]
            blocked = j >= torch.maximum(torch.tensor(S, device=qkv.device), i + 1)
            s = s.masked_fill(blocked[None], float("-inf"))
        o = torch.softmax(s, -1) @ v
        outs.append(o.transpose(0, 1).reshape(L, d))
        off += L
    return torch.cat(outs, 0)


@pytest.mark.parametrize("nhead,dh", [(16, 4), (4, 16), (4, 32), (2, 64), (16, 64), (2, 96)])
@pytest.mark.parametrize("causal", [True, False])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_attention_rows(nhead, dh, causal, dt):
    d = nhead * dh
    lens = [70, 1, 133, 64]
    text_lens = [9, 1, 20, 64]
    rows = sum(lens)
    qkv = (_rand(rows, 3 * d, seed=14)).to(dt)
    so = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=DEV)
    tl = torch.tensor(text_lens, dtype=torch.int32, device=DEV)
    out = ops.attention(qkv, so, tl, nhead, causal)
    ref = _ref_attention(qkv, lens, text_lens, nhead, causal)
    err = (out.double() - ref).abs().max().item()
    assert err < (2e-5 if dt == torch.float32 else 0.02), err


@pytest.mark.parametrize("nhead,dh", [(16, 64), (2, 96), (4, 16), (8, 128)])
@pytest.mark.parametrize("causal", [True, False])
def test_fp32_attention_vector_staging_is_bit_identical(nhead, dh, causal):
    """attention.hip, the token-exact mode's attention (round 6 knob "attn_f32_vec"): 16-byte, register-double-buffered K / V staging
    changes how tiles reach LDS -- not what a row computes: the same bits as round 1's element-wise staging, on ragged packed sequences
    incl. M = 1025 (fp32 NAR 55.2 -> 49.1 ms, profiles/r06_fp32_glds.json)."""
    d = nhead * dh
    lens, text_lens = [1025, 70, 1, 133], [47, 9, 1, 20]
    qkv = _rand(sum(lens), 3 * d, seed=15)
    so = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=DEV)
    tl = torch.tensor(text_lens, dtype=torch.int32, device=DEV)
    try:
        ops.tune("attn_f32_vec", 0)
        old = ops.attention(qkv, so, tl, nhead, causal).clone()
    finally:
        ops.tune("attn_f32_vec", 1)
    new = ops.attention(qkv, so, tl, nhead, causal)
    assert torch.equal(old, new)
    ref = _ref_attention(qkv, lens, text_lens, nhead, causal)
    assert (new.double() - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("nhead,dh,lens,text_lens", [
    (16, 64, [1025], [47]), (16, 64, [272, 300, 65], [47, 100, 64]), (8, 128, [200, 129], [30, 129]), (4, 32, [513], [1]),
    (2, 96, [1100], [64]),
])
@pytest.mark.parametrize("causal", [True, False])
@pytest.mark.parametrize("v2,qw,mode,q128,defer", [(2, 0, 3, 0, 8), (2, 0, 3, 1, 0), (2, 0, 2, 0, 8), (2, 0, 2, 1, 0), (2, 0, 1, 0, 8), (2, 0, 0, 1, 8), (2, 0, 1, 0, 0), (0, 1, 1, 0, 8),
                                                      (0, 2, 0, 0, 8)])
def test_attention_mfma_long(nhead, dh, lens, text_lens, causal, v2, qw, mode, q128, defer):
    """bf16 MFMA flash kernels at NAR / prefill lengths (many key tiles, ragged tails, prefix-LM mask): the second-generation
    kernel (attn_mfma2.hip: double-buffered tiles; tile staging by LDS-DMA + XOR swizzle with a V^T pre-pass (mode 2) or with V
    row-major and LDS transpose reads (mode 3) (head sizes 64 / 128) or through registers with 32- / 16-byte row padding (mode 1 / 0); 64- and 128-query blocks; deferred (8) or exact (0)
    running maximum) and round 1's (attn_mfma.hip) with 64- and 128-query blocks (knobs attn_v2 / attn_qw / attn_mode / attn_q128 /
    attn_defer)."""
    ops.tune("attn_v2", v2)
    ops.tune("attn_qw", qw)
    ops.tune("attn_mode", mode)
    ops.tune("attn_q128", q128)
    ops.tune("attn_defer", defer)
    try:
        for ring in ((2, 4) if (v2 == 2 and mode == 2) else (0,)):  # (mode 3 = mode 2 with V read through ds_read_b64_tr_b16)  # LDS ring depth of the LDS-DMA staging (0 = by launch size)
            ops.tune("attn_ring", ring)
            for lsum in ((1, 0) if mode == 3 else (1,)):  # mode 3: softmax row sums on the MFMA pipe (default) / VALU adds
                ops.tune("attn_lsum", lsum)
                _attention_mfma_long(nhead, dh, lens, text_lens, causal)
    finally:
        ops.tune("attn_lsum", 1)
        ops.tune("attn_ring", 0)
        ops.tune("attn_qw", 0)
        ops.tune("attn_v2", 1)
        ops.tune("attn_mode", 3)
        ops.tune("attn_q128", -1)
        ops.tune("attn_defer", 8)


def _attention_mfma_long(nhead, dh, lens, text_lens, causal):
    d = nhead * dh
    rows = sum(lens)
    qkv = (_rand(rows, 3 * d, seed=15)).to(torch.bfloat16)
    so = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=DEV)
    tl = torch.tensor(text_lens, dtype=torch.int32, device=DEV)
    out = ops.attention(qkv, so, tl, nhead, causal)
    ref = _ref_attention(qkv, lens, text_lens, nhead, causal)
    err = (out.double() - ref).abs().max().item()
    assert err < 0.02, err
    # spiky scores: one key dominates (exercises the running-max rescale)
    qkv2 = qkv.clone()
    qkv2[:, d : 2 * d] *= 6.0
    out = ops.attention(qkv2, so, tl, nhead, causal)
    ref = _ref_attention(qkv2, lens, text_lens, nhead, causal)
    assert (out.double() - ref).abs().max().item() < 0.03


def _ref_decode(q, kc, vc, kv_len):
    B, H, ctx_max, dh = kc.shape
    outs = []
    for b in range(B):
        n = int(kv_len[b]) + 1
        qq = q[b].double().view(H, 1, dh)
        s = qq @ kc[b, :, :n].double().transpose(1, 2) / math.sqrt(dh)
        outs.append((torch.softmax(s, -1) @ vc[b, :, :n].double()).reshape(H * dh))
    return torch.stack(outs)


# (H, dh, ctx_max, kv_len per utterance): single chunk, many rounds, context = 1, full cache, ragged batch
DEC_CASES = [
    (16, 64, 1026, [271]), (16, 64, 1026, [1025]), (16, 64, 1026, [0]), (16, 64, 1026, [127]), (16, 64, 1026, [128]),
    (4, 64, 300, [5, 299, 150]), (16, 4, 200, [199, 3]), (4, 16, 520, [519]), (2, 96, 700, [650, 17]), (4, 32, 90, [89]),
    (16, 64, 2100, [2099, 1000]),
]


@pytest.mark.parametrize("H,dh,ctx_max,kv", DEC_CASES)
@pytest.mark.parametrize("nsplit", [1, 4, 8, 16])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_decode_attention(H, dh, ctx_max, kv, nsplit, dt):
    B, d = len(kv), H * dh
    q = _rand(B, d, seed=20)
    kc = _rand(B, H, ctx_max, dh, seed=21).to(dt)
    vc = _rand(B, H, ctx_max, dh, seed=22).to(dt)
    # slots beyond the context hold stale but finite data in the engine; make them adversarially large
    for b in range(B):
        kc[b, :, kv[b] + 1 :] = 50.0
        vc[b, :, kv[b] + 1 :] = -1000.0
    kl = torch.tensor(kv, dtype=torch.int32, device=DEV)
    out, ws = ops.decode_attention(q, kc, vc, kl, nsplit=nsplit)
    ref = _ref_decode(q, kc, vc, kv)
    err = (out.double() - ref).abs().max().item()
    assert err < 2e-5, err  # K/V are exact in fp32 for both dtypes; only fp32 roundoff + fast exp remain
    # out_proj with the split merge fused in its prologue (batch <= 8)
    w = (_rand(d, d, seed=23) / math.sqrt(d)).to(dt)
    bias = _rand(d, seed=24) * 0.1
    r0 = _rand(B, d, seed=25)
    got = ops.attn_out_proj(ws, w, bias, r0.clone(), H, nsplit)
    want = r0.double() + ref @ w.double().t() + bias.double()
    assert (got.double() - want).abs().max().item() < 1e-4


# (d, H, dtype): every (chunks per row, head size) instantiation of the fused launch -- fp32 d256 (1 chunk), d1024 (4 chunks: two
# query-row passes per wave); bf16 d512 / d1024 / d2048; head size 128
STEP1_SHAPES = [(256, 4, torch.float32), (1024, 16, torch.float32), (512, 8, torch.bfloat16), (1024, 16, torch.bfloat16),
                (2048, 16, torch.bfloat16), (1024, 8, torch.bfloat16), (512, 4, torch.float32)]


@pytest.mark.parametrize("d,H,dt", STEP1_SHAPES)
@pytest.mark.parametrize("nsplit", [4, 8, 16])
@pytest.mark.parametrize("kv_len,ctx_max", [(0, 40), (1, 40), (271, 1026), (127, 1026), (128, 1026), (129, 1026), (1025, 1026), (2050, 2100)])
@pytest.mark.parametrize("waves", [8, 4])
def test_fused_qkv_attention_step_vs_fp64(d, H, dt, nsplit, kv_len, ctx_max, waves):
    """vle_op_attn_step1 = the batch-1 AR step's two attention launches (qkv_attn1_kernel: LN1 + QKV GEMV + cache append + decode
    attention over the OLD keys; out-proj GEMV with the new token's own softmax term merged in its prologue) against the fp64
    definition  x + out_proj(softmax(q K^T / sqrt(dh)) V)  with K / V = the cache rows 0 .. kv_len-1 plus the new token's
    (rounded to the cache type).  Also: the cache slot kv_len holds exactly the rounded K / V, no other slot changes."""
    if nsplit == 16 and d // 64 > 16:
        pytest.skip("16 splits x 4 chunks of partials do not fit the out-proj GEMV's registers (the engine never asks)")
    dh = d // H
    g = torch.Generator().manual_seed(d + kv_len + nsplit)
    x = torch.randn(d, generator=g) * 1.5 + 0.3
    gamma, beta = torch.rand(d, generator=g) + 0.5, torch.randn(d, generator=g) * 0.1
    w_in = (torch.randn(3 * d, d, generator=g) / math.sqrt(d)).to(dt)
    b_in = torch.randn(3 * d, generator=g) * 0.1
    w_out = (torch.randn(d, d, generator=g) / math.sqrt(d)).to(dt)
    b_out = torch.randn(d, generator=g) * 0.1
    kc = torch.randn(H, ctx_max, dh, generator=g).to(dt)
    vc = torch.randn(H, ctx_max, dh, generator=g).to(dt)
    kc[:, kv_len:] = 50.0   # stale slots: finite, adversarially large
    vc[:, kv_len:] = -1000.0
    kc_d, vc_d = kc.clone().to(DEV), vc.clone().to(DEV)
    ops.tune("qa_waves", waves)  # 4 = default; 8 falls back to 4 where a 512-thread workgroup would leave threads without a row element
    try:
        got = ops.attn_step1(x.to(DEV), gamma.to(DEV), beta.to(DEV), w_in.to(DEV), b_in.to(DEV), w_out.to(DEV), b_out.to(DEV), kc_d, vc_d,
                             kv_len, H, nsplit).cpu()
    finally:
        ops.tune("qa_waves", 4)
    xd = x.double()
    xn = (xd - xd.mean()) / torch.sqrt(xd.var(unbiased=False) + 1e-5) * gamma.double() + beta.double()
    qkv = w_in.double() @ xn + b_in.double()
    q, kn, vn = qkv[:d].view(H, dh), qkv[d: 2 * d].view(H, dh), qkv[2 * d:].view(H, dh)
    kn_r, vn_r = kn.to(dt).double(), vn.to(dt).double()
    K = torch.cat([kc[:, :kv_len].double(), kn_r[:, None]], dim=1)  # (H, kv_len + 1, dh)
    V = torch.cat([vc[:, :kv_len].double(), vn_r[:, None]], dim=1)
    p = torch.softmax(torch.einsum("hd,hkd->hk", q, K) / math.sqrt(dh), dim=-1)
    att = torch.einsum("hk,hkd->hd", p, V).reshape(d)
    want = xd + w_out.double() @ att + b_out.double()
    tol = 2e-4 if dt == torch.float32 else 2e-2  # bf16: the cache-rounding of the new K / V sits on a rounding boundary now and then
    assert (got.double() - want).abs().max().item() < tol * max(1.0, want.abs().max().item()), (got.double() - want).abs().max().item()
    # cache append: slot kv_len = the token's K / V in the cache type (one rounding of the fp32 GEMV result), nothing else touched
    kslot, vslot = kc_d[:, kv_len].cpu().double(), vc_d[:, kv_len].cpu().double()
    ulp = 1e-5 if dt == torch.float32 else 2.0 ** -7
    assert (kslot - kn).abs().max().item() <= ulp * max(1.0, kn.abs().max().item())
    assert (vslot - vn).abs().max().item() <= ulp * max(1.0, vn.abs().max().item())
    keep = torch.ones(ctx_max, dtype=torch.bool)
    keep[kv_len] = False
    assert torch.equal(kc_d[:, keep].cpu(), kc[:, keep]) and torch.equal(vc_d[:, keep].cpu(), vc[:, keep])


def test_fused_qkv_attention_step_equals_the_three_launch_path():
    """The fused pair against the engine's previous three launches (QKV GEMV with cache write, decode attention over slots
    0 .. kv_len, out-proj with the split merge) on the same operands: equal up to fp32 summation order."""
    d, H, dt = 1024, 16, torch.bfloat16
    dh, ctx_max, kv_len = d // H, 1026, 600
    g = torch.Generator().manual_seed(9)
    x = torch.randn(d, generator=g)
    gamma, beta = torch.rand(d, generator=g) + 0.5, torch.randn(d, generator=g) * 0.1
    w_in = (torch.randn(3 * d, d, generator=g) / math.sqrt(d)).to(dt).to(DEV)
    b_in = (torch.randn(3 * d, generator=g) * 0.1).to(DEV)
    w_out = (torch.randn(d, d, generator=g) / math.sqrt(d)).to(dt).to(DEV)
    b_out = (torch.randn(d, generator=g) * 0.1).to(DEV)
    kc = torch.randn(H, ctx_max, dh, generator=g).to(dt).to(DEV)
    vc = torch.randn(H, ctx_max, dh, generator=g).to(dt).to(DEV)
    k1, v1 = kc.clone(), vc.clone()
    fused = ops.attn_step1(x.to(DEV), gamma.to(DEV), beta.to(DEV), w_in, b_in, w_out, b_out, k1, v1, kv_len, H, 8)
    # reference path from stand-alone operators: q / k / v by the LayerNorm-fused GEMV, cache append by hand
    qkv = ops.linear_skinny(x[None].to(DEV), w_in, b_in, epilogue=0, gamma=gamma.to(DEV), beta=beta.to(DEV))[0]
    k2, v2 = kc.clone(), vc.clone()
    k2[:, kv_len] = qkv[d: 2 * d].view(H, dh).to(dt)
    v2[:, kv_len] = qkv[2 * d:].view(H, dh).to(dt)
    assert torch.equal(k1, k2) and torch.equal(v1, v2)
    _, ws = ops.decode_attention(qkv[None, :d].contiguous(), k2[None], v2[None], torch.tensor([kv_len], dtype=torch.int32, device=DEV), nsplit=4)
    want = ops.attn_out_proj(ws, w_out, b_out, x[None].to(DEV).clone(), H, 4)[0]
    assert (fused - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("M,N,K", [(64, 1024, 4096), (64, 1024, 1024), (33, 1025, 1024), (5, 3072, 1024), (64, 1536, 1536)])
@pytest.mark.parametrize("ksplit", [None, 1, 2, 4, 8])
@pytest.mark.parametrize("epi", [ops.EPI_RESID, ops.EPI_F32, ops.EPI_RELU])
def test_linear_split_k_tickets(M, N, K, ksplit, epi):
    """gemm_skinny split-K: every slice count gives the fp64 product within bf16-input roundoff, the result
    does not depend on arrival order (bit-identical across repeats) and the tickets reset themselves."""
    if ksplit and K % (256 * ksplit):
        pytest.skip("slice would be shorter than one wave chunk")
    a = _rand(M, K, seed=40).to(torch.bfloat16)
    w = (_rand(N, K, seed=41) / math.sqrt(K)).to(torch.bfloat16)
    bias = _rand(N, seed=42) * 0.1
    ref = a.double() @ w.double().t() + bias.double()
    r0 = _rand(M, N, seed=43)
    outs = []
    for rep in range(3):
        if epi == ops.EPI_RESID:
            o = ops.linear(a, w, bias, epi, resid=r0.clone(), ksplit=ksplit)
        else:
            o = ops.linear(a, w, bias, epi, ksplit=ksplit)
        outs.append(o.clone())
    want = ref + r0.double() if epi == ops.EPI_RESID else ref.clamp_min(0) if epi == ops.EPI_RELU else ref
    err = (outs[0].double() - want).abs().max().item()
    tol = 2e-5 * math.sqrt(K / 64) if epi != ops.EPI_RELU else 0.02  # RELU stores bf16
    assert err < tol, (err, tol)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    if ksplit is not None:
        torch.cuda.synchronize()
        assert int(ops.linear_workspace(a.device)[:4096].view(torch.int32).abs().sum()) == 0, "tickets not reset"


@pytest.mark.parametrize("M,N,K", [(64, 1024, 4096), (64, 1024, 1024), (33, 1025, 1024), (24, 1536, 6144), (48, 1024, 1536), (17, 512, 2048), (64, 1536, 1536)])
@pytest.mark.parametrize("epi", [ops.EPI_RESID, ops.EPI_F32, ops.EPI_RELU])
@pytest.mark.parametrize("fp8w", [False, True])
def test_skinny_gemm_m_split_kernel(M, N, K, epi, fp8w):
    """gemm_skinny.hip's M-split kernel (one workgroup per (16 rows of W, 16 utterances), all of K, 4 / 8 / 16 waves; taken by
    the engine's `ksplit = 0` policy where N / 16 row fragments alone would leave CUs idle) against the fp64 product and
    against the split-K kernel it replaces (knob gs_msplit = 0): same values up to fp32 summation order, deterministic."""
    a = _rand(M, K, seed=140).to(torch.bfloat16)
    w32 = _rand(N, K, seed=141) / math.sqrt(K)
    bias = _rand(N, seed=142) * 0.1
    r0 = _rand(M, N, seed=143)
    if fp8w:
        from oracle import valle_oracle as vo
        q8, sc, deq = vo.fp8w_quantize(w32.cpu())
        q8, sc, wref = q8.to(DEV), sc.to(DEV), deq.to(DEV)
    else:
        wbf = w32.to(torch.bfloat16)
        wref = wbf.float()

    def run():
        kw = dict(resid=r0.clone()) if epi == ops.EPI_RESID else {}
        if fp8w:
            return ops.linear_fp8w(a, q8, sc, bias, epi, ksplit=0, **kw).clone()
        return ops.linear(a, wbf, bias, epi, ksplit=0, **kw).clone()

    ref = a.double() @ wref.double().t() + bias.double()
    want = ref + r0.double() if epi == ops.EPI_RESID else ref.clamp_min(0) if epi == ops.EPI_RELU else ref
    tol = 2e-5 * math.sqrt(K / 64) * max(1.0, want.abs().max().item()) if epi != ops.EPI_RELU else 0.02 * max(1.0, want.abs().max().item())
    ops.tune("gs_msplit", 2)  # every K (the default, 1, keeps K > 2048 on the split-K kernel)
    try:
        outs = [run(), run()]
        assert torch.equal(outs[0], outs[1])
        assert (outs[0].double() - want).abs().max().item() < tol
        ops.tune("gs_msplit", 3)  # non-temporal W loads
        assert torch.equal(run(), outs[0])
        ops.tune("gs_msplit", 0)
        old = run()
    finally:
        ops.tune("gs_msplit", 1)
    assert (old.double() - want).abs().max().item() < tol
    assert (outs[0].double() - old.double()).abs().max().item() < tol


@pytest.mark.parametrize("M,N,K", [(17408, 1024, 1024), (1041, 4096, 1024), (300, 1025, 1024), (130, 256, 256), (4100, 3072, 1536), (2049, 1024, 4096)])
@pytest.mark.parametrize("knob", ["glds_swz", "glds_prio", "glds_epi", "glds_swz+glds_epi", "glds_8ph", "glds_8ph+g8_stagger", "glds_8ph+glds_swz", "glds_8ph+g8_stagger+glds_swz", "glds_8ph+g8_stagger+g8_colgroup", "glds_8ph+g8_stagger+g8_dbg"])  # listed = 1 (g8_colgroup: 4; g8_dbg: 4 = stores straight from the fragments instead of the LDS-staged epilogue), others 0
def test_linear_gemm_tile_policy_knobs(M, N, K, knob):
    """The A/B variants of the 8-wave GEMM tiles (alternative LDS slot key, s_setprio) compute the same product."""
    a = _rand(M, K, seed=70).to(torch.bfloat16)
    w = (_rand(N, K, seed=71) / math.sqrt(K)).to(torch.bfloat16)
    bias = _rand(N, seed=72) * 0.1
    ref = a.double() @ w.double().t() + bias.double()
    knobs = knob.split("+")
    # glds_epi: 1 (default) = gemm_glds.hip's epilogue staged through LDS, 0 = stores straight from the fragments -- the variants
    # that do not list it run the legacy epilogue and must still equal `base` (staged) bit for bit
    defaults = {"glds_swz": 0, "glds_prio": 0, "glds_epi": 1, "glds_8ph": -1, "g8_stagger": 1, "g8_colgroup": 0, "g8_dbg": 0}

    def tune(on):
        for k, dflt in defaults.items():
            ops.tune(k, ((4 if k in ("g8_colgroup", "g8_dbg") else 1) if k in knobs else 0) if on else dflt)

    ops.tune("glds_8ph", 0)
    base = ops.linear(a, w, bias, ops.EPI_F32, ksplit=None)  # the 8-wave tile kernels of gemm_glds.hip

    tune(1)
    try:
        got = ops.linear(a, w, bias, ops.EPI_F32, ksplit=None)
    finally:
        tune(0)
    assert (got.double() - ref).abs().max().item() < 3e-5 * math.sqrt(K / 64) * max(1.0, ref.abs().max().item())
    if "glds_8ph" not in knobs:
        assert torch.equal(got, base)  # same MFMA order, only the LDS placement / issue priority differs
    else:  # other tile shape: also the store / ReLU / residual epilogues, and repeat runs must agree bit for bit (race screen)
        r0 = _rand(M, N, seed=73)
        tune(1)
        try:
            outs = [ops.linear(a, w, bias, ops.EPI_RESID, resid=r0.clone(), ksplit=None) for _ in range(4)]
            relu = ops.linear(a, w, bias, ops.EPI_RELU, ksplit=None)
        finally:
            tune(0)
        assert all(torch.equal(outs[0], o) for o in outs[1:])
        assert (outs[0].double() - (ref + r0.double())).abs().max().item() < 3e-5 * math.sqrt(K / 64) * max(1.0, ref.abs().max().item())
        assert (relu.double() - ref.clamp_min(0)).abs().max().item() < 0.02 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("M,N,K", [(64 * 256 + 64, 1024, 1024), (64 * 256 + 37, 1024, 4096), (32 * 256 + 200, 2048, 1024), (21 * 256 + 1, 3072, 1024),
                                   (1025, 4096, 1024), (2 * 1024 + 2, 2048, 1024)])  # the last two: the 128 x 128 tiles (one / two utterances' linear1)
def test_linear_gemm_leftover_rows_as_a_second_launch(M, N, K):
    """Knob "glds_tail" (gemm_glds.hip): the rows past the last full 256-row tile of a 256 x 256-tile launch run as a second, small
    launch when the ragged tile row would cost a whole extra round over the 256 CUs (the NAR stages' M = 1024 B + B).  Rows are
    independent: with and without the split every epilogue gives the same bits, and the leftover rows are right."""
    a = _rand(M, K, seed=80).to(torch.bfloat16)
    w = (_rand(N, K, seed=81) / math.sqrt(K)).to(torch.bfloat16)
    bias = _rand(N, seed=82) * 0.1
    r0 = _rand(M, N, seed=83)
    ref = a.double() @ w.double().t() + bias.double()
    if (M // 256) * (N // 256) >= 128:
        assert -(-(M // 256) * (N // 256) // 256) < -(-((M + 255) // 256) * (N // 256) // 256), "a shape the split applies to"
    else:
        assert 0 < M % 128 <= 64 and -(-(M // 128) * (N // 128) // 256) < -(-((M + 127) // 128) * (N // 128) // 256), "a shape the split applies to"

    def run():
        return (ops.linear(a, w, bias, ops.EPI_F32, ksplit=None), ops.linear(a, w, bias, ops.EPI_STORE, ksplit=None),
                ops.linear(a, w, bias, ops.EPI_RELU, ksplit=None), ops.linear(a, w, bias, ops.EPI_RESID, resid=r0.clone(), ksplit=None))

    ops.tune("glds_tail", 0)
    try:
        whole = run()
    finally:
        ops.tune("glds_tail", 1)
    split = run()
    for x, y in zip(whole, split):
        assert torch.equal(x, y)
    tol = 3e-5 * math.sqrt(K / 64) * max(1.0, ref.abs().max().item())
    assert (split[0].double() - ref).abs().max().item() < tol
    assert (split[3].double() - (ref + r0.double())).abs().max().item() < tol
    tail = slice(M - M % (256 if (M // 256) * (N // 256) >= 128 else 128), M)
    assert (split[1][tail].double() - ref[tail]).abs().max().item() < 0.02 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("M,N,K", [(1025, 3072, 1024), (1025, 1024, 4096), (1025, 4096, 1024), (272, 1024, 1024), (128, 64, 32), (2050, 1536, 1536), (131, 260, 96)])
def test_fp32_gemm_on_the_lds_dma_ring_is_bit_identical_to_the_register_staged_kernel(M, N, K):
    """Round 6 ("f32_glds"): the token-exact mode's packed-row GEMMs (M >= 128) run gemm_glds.hip's LDS-DMA ring instantiated for fp32
    operands -- 64 x 64 tiles, v_mfma_f32_16x16x4_f32 fed the same 16-byte fragments in the same stage / half / component order as
    gemm.hip, i.e. the same chain of exact fp32 FMAs per output element.  Every epilogue must give the SAME BITS as gemm.hip (knob off), so
    the reference's greedy ids (tests/test_parity_sizes_gpu.py) cannot move; also against the fp64 product, ragged M, N not a tile multiple."""
    a = _rand(M, K, seed=180)
    w = _rand(N, K, seed=181) / math.sqrt(K)
    bias = _rand(N, seed=182) * 0.1
    r0 = _rand(M, N, seed=183)

    def run():
        return [ops.linear(a, w, bias, ops.EPI_F32, ksplit=None), ops.linear(a, w, bias, ops.EPI_STORE, ksplit=None),
                ops.linear(a, w, bias, ops.EPI_RELU, ksplit=None), ops.linear(a, w, bias, ops.EPI_RESID, resid=r0.clone(), ksplit=None),
                ops.linear(a, w, None, ops.EPI_STORE, ksplit=None)]

    ops.tune("f32_glds", 0)
    try:
        old = run()
    finally:
        ops.tune("f32_glds", 1)
    new, again = run(), run()
    for i, (x, y, z) in enumerate(zip(old, new, again)):
        assert x.dtype == torch.float32 and torch.equal(x, y), (i, (x - y).abs().max().item())
        assert torch.equal(y, z), (i, "run-to-run")
    ref = a.double() @ w.double().t() + bias.double()
    assert (new[0].double() - ref).abs().max().item() < 3e-5 * math.sqrt(K / 64) * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("M,N,K", [(64 * 256, 1024, 1024), (65600, 1024, 4096), (65600, 3072, 1024), (40 * 256 + 37, 4096, 1024), (33 * 256, 1024, 256),
                                   (150 * 256 + 255, 512, 1536)])
def test_gemm_8ph_persistent_tile_loop_is_bit_identical(M, N, K):
    """Knob "g8_persist" (gemm_8ph.hip, round 6): one workgroup per CU walks several 256 x 256 tiles -- the DMA queue runs across
    tiles, the epilogue stages through one LDS buffer while the other receives the next tile's first K-tile, its stores are never
    waited for.  Arithmetic, reduction orders and roundings are the one-tile kernel's: every epilogue (fp32, bf16, ReLU, residual
    read-modify-write, and the LayerNorm-folded producer / consumer pair) must give the SAME BITS with the knob on and off -- with
    1 / 2 / 4+ tiles per workgroup, ragged last tile rows (with glds_tail = 0 they stay inside the launch), K = 256 (four K-tiles: the
    cross-tile requests start in the first trip) -- and repeated runs must agree (race screen)."""
    assert (M // 256) * (N // 256) >= 128, "a shape the 256 x 256 tiles take"
    a = _rand(M, K, seed=170).to(torch.bfloat16)
    w = (_rand(N, K, seed=171) / math.sqrt(K)).to(torch.bfloat16)
    bias = _rand(N, seed=172) * 0.1
    r0 = _rand(M, N, seed=173)
    d = N
    gamma = 1.0 + 0.3 * _rand(d, seed=174)
    ln_ok = d % 256 == 0 and d <= 1536

    def run():
        outs = [ops.linear(a, w, bias, ops.EPI_F32, ksplit=None), ops.linear(a, w, bias, ops.EPI_STORE, ksplit=None),
                ops.linear(a, w, bias, ops.EPI_RELU, ksplit=None), ops.linear(a, w, bias, ops.EPI_RESID, resid=r0.clone(), ksplit=None)]
        if ln_ok:  # producer (this GEMM completes the residual stream of width N), then a consumer of its xg / statistics
            x = r0.clone()
            xg, stats = ops.linear_ln_producer(a, w, bias, x, gamma)
            wc = (_rand(2 * d, d, seed=175) / math.sqrt(d)).to(torch.bfloat16)
            sg, tb = _rand(2 * d, seed=176), _rand(2 * d, seed=177)
            outs += [x, xg, stats, ops.linear_ln_consumer(xg, wc, tb, sg, stats, relu=False), ops.linear_ln_consumer(xg, wc, tb, sg, stats, relu=True)]
        torch.cuda.synchronize()
        return outs

    for tail in (1, 0):  # 0: a ragged last tile row stays inside the 256 x 256 launch (rows clamped on load, dropped on store)
        ops.tune("glds_tail", tail)
        try:
            ops.tune("g8_persist", 0)
            one = run()
            ops.tune("g8_persist", 7)  # every epilogue class on the persistent loop (the engine's default policy is a subset)
            loop = run()
            again = run()
        finally:
            ops.tune("g8_persist", 1)
            ops.tune("glds_tail", 1)
        for i, (x, y, z) in enumerate(zip(one, loop, again)):
            assert torch.equal(x, y), (tail, i, (x.double() - y.double()).abs().max().item())
            assert torch.equal(y, z), (tail, i, "run-to-run")
    ref = a.double() @ w.double().t() + bias.double()
    assert (loop[0].double() - ref).abs().max().item() < 3e-5 * math.sqrt(K / 64) * max(1.0, ref.abs().max().item())


# (M, d): every tile policy of gemm_glds.hip / gemm_8ph.hip the prefill / NAR rows meet -- 64 x 64 (few rows), 128 x 64 / 128 x 128 with the
# 64 x 64 leftover launch (one utterance, M = 1025), 256 x 128, 256 x 256 phase-split + leftover rows (batched NAR rows), d 1536 (24 groups)
@pytest.mark.parametrize("M,d", [(128, 256), (272, 1024), (1025, 1024), (2 * 1025, 1024), (8 * 1025, 1024), (33 * 1025, 1024), (1025, 1536), (12 * 1025, 1536)])
def test_layernorm_folded_into_the_gemms_equals_layernorm_then_linear(M, d):
    """kernels.h GemmLn: the GEMM that completes the residual stream (out-proj / linear2, here K = d and K = 4 d) also leaves
    xg = bf16(x * gamma) and per-64-column-group (mean, M2); the GEMM that reads the normalised row (in-proj / linear1) multiplies xg
    and applies rstd * (acc - mean * sg) + tb.  Against the fp64 definition LayerNorm(x) W^T + b of valle/modules/transformer.py:57-74
    + F.linear (biased variance, eps 1e-5), with a row mean several times the row's spread (the fold's cancellation case)."""
    g = torch.Generator().manual_seed(5 + M + d)
    for Kp in (d, 4 * d):  # producer K: out-proj, linear2
        a = (torch.randn(M, Kp, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
        wp = (torch.randn(d, Kp, generator=g) / math.sqrt(Kp)).to(torch.bfloat16).to(DEV)
        bp = (torch.randn(d, generator=g) * 0.1).to(DEV)
        x0 = (torch.randn(M, d, generator=g) * 1.5 + torch.randn(M, 1, generator=g) * 2.0).to(DEV)  # row means up to several sigma
        gamma = (1.0 + 0.3 * torch.randn(d, generator=g)).to(DEV)
        beta = (0.2 * torch.randn(d, generator=g)).to(DEV)
        x_ref = x0.double() + a.double() @ wp.double().t() + bp.double()
        x = x0.clone()
        xg, stats = ops.linear_ln_producer(a, wp, bp, x, gamma)
        torch.cuda.synchronize()
        tol = 3e-5 * math.sqrt(Kp / 64) * max(1.0, x_ref.abs().max().item())
        assert (x.double() - x_ref).abs().max().item() < tol, "residual stream"
        # the side products are functions of the fp32 row the kernel stored: exact definitions on x itself
        assert torch.equal(xg, (x * gamma).to(torch.bfloat16)), "bf16(x * gamma)"
        xg64 = x.double().view(M, d // 64, 64)
        gm = xg64.mean(-1)
        gm2 = ((xg64 - gm[..., None]) ** 2).sum(-1)
        st = stats.permute(1, 0, 2)  # (M, groups, 2): the operator's layout is group-major
        assert (st[..., 0].double() - gm).abs().max().item() < 1e-5 * max(1.0, gm.abs().max().item())
        assert ((st[..., 1].double() - gm2).abs() / gm2.clamp_min(1e-3)).max().item() < 1e-4
        for N, relu in ((3 * d, False), (4 * d, True)):  # consumers: in-projection, linear1 + ReLU
            wc = (torch.randn(N, d, generator=g) / math.sqrt(d)).to(torch.bfloat16).to(DEV)
            bc = (torch.randn(N, generator=g) * 0.1).to(DEV)
            sg = (wc.double() @ gamma.double()).float()
            tb = (wc.double() @ beta.double() + bc.double()).float()
            out = ops.linear_ln_consumer(xg, wc, tb, sg, stats, relu=relu)
            mean = x.double().mean(-1, keepdim=True)
            var = ((x.double() - mean) ** 2).mean(-1, keepdim=True)
            ln = (x.double() - mean) / torch.sqrt(var + 1e-5) * gamma.double() + beta.double()
            ref = ln @ wc.double().t() + bc.double()
            if relu:
                ref = ref.clamp_min(0)
            # what the un-folded path computes: bf16(LN(x)) @ W -- the fold must sit as close to the definition as that does
            unf = ln.to(torch.bfloat16).double() @ wc.double().t() + bc.double()
            if relu:
                unf = unf.clamp_min(0)
            scale = max(1.0, ref.abs().max().item())
            e_fold = (out.double() - ref).abs().max().item() / scale
            e_unf = (unf.to(torch.bfloat16).double() - ref).abs().max().item() / scale
            assert e_fold < max(3 * e_unf, 0.02), (M, d, Kp, N, e_fold, e_unf)
            assert (out.double() - ref).abs().mean().item() / scale < 2e-3


@pytest.mark.parametrize("formal", [0, 1])
def test_skinny_gemm_split_k_handoff_under_memory_pressure(formal):
    """`formal` = 1: the same hand-off with explicit agent-scope release / acquire fences around the ticket (knob "gs_formal").
    Stress of the cross-workgroup split-K hand-off of gemm_skinny.hip (write-through partial tiles + vmcnt drain + relaxed
    ticket, no fences): 150 launches at 2 / 4 / 8 K slices while a second stream saturates HBM with copies, so workgroup
    arrival order and cache state vary from launch to launch; every result must be bit-identical to the first one of its slice
    count (the combine sums in slice order), and equal to the un-split kernel up to fp32 summation order."""
    torch.manual_seed(3)
    M, N, K = 64, 1024, 4096
    a = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    w = (torch.randn(N, K, device=DEV) / math.sqrt(K)).to(torch.bfloat16)
    bias = torch.randn(N, device=DEV)
    ref = ops.linear(a, w, bias, ops.EPI_F32, ksplit=1)
    side = torch.cuda.Stream()
    big = torch.empty(64 << 20, dtype=torch.float32, device=DEV)
    big2 = torch.empty_like(big)
    ops.tune("gs_formal", formal)
    try:
        for ks in (2, 4, 8):
            first = ops.linear(a, w, bias, ops.EPI_F32, ksplit=ks).clone()
            assert (first - ref).abs().max().item() <= 1e-3 * ref.abs().max().item()
            for it in range(50):
                with torch.cuda.stream(side):
                    big2.copy_(big)
                out = ops.linear(a, w, bias, ops.EPI_F32, ksplit=ks)
                assert torch.equal(out, first), f"split-K x{ks}: launch {it} differs"
        torch.cuda.synchronize()
    finally:
        ops.tune("gs_formal", 0)
