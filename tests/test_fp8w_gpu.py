"""GPU parity of engine mode FP8W (fp8 e4m3fn weight rows with one power-of-two scale each).

The mode is DEFINED as "the bf16 engine on W' = e4m3fn(w / 2^e) * 2^e": W' is exact in bf16 and 2^e is an exact
factor, so (a) the batch-1 GEMV on the fp8 codes must be BIT-identical to the bf16 GEMV on bf16(W') (the batch GEMM
only differs in fp32 summation order),
(b) a whole batch-1 decode in FP8W mode must equal, token for token and logit for logit, the bf16 engine
loaded with W', and (c) against the CPU oracle run on W' (oracle.fp8w_state_dict) the usual bf16 tolerance
applies (teacher-forced, max|dlogit| <= 5 % sigma)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import valle_amd  # noqa: E402
from valle_amd import ops  # noqa: E402
from oracle import valle_oracle as vo  # noqa: E402
from tests.test_engine_gpu import build_model  # noqa: E402

DEV = "cuda:0"


def _rand(*shape, seed=0, scale=1.0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale).to(DEV)


def _quant(N, K, seed):
    w = torch.randn(N, K, generator=torch.Generator().manual_seed(seed)) / math.sqrt(K)
    w = w * torch.logspace(-1, 1, N)[:, None]  # rows of very different magnitude: the scales matter
    q, s, deq = ops.quantize_fp8w(w)
    return q.to(DEV), s.to(DEV), deq.to(DEV)


def test_device_fp8_decode_of_all_codes():
    """v_cvt_pk_f32_fp8 on gfx950 must read OCP e4m3fn: a GEMV with x = e_k returns column k of W'."""
    codes = torch.arange(256, dtype=torch.uint8)
    want = codes.view(torch.float8_e4m3fn).to(torch.float32)
    finite = ~torch.isnan(want)
    codes = torch.where(finite, codes, torch.zeros((), dtype=torch.uint8))
    K = 512
    w8 = torch.zeros(256, K, dtype=torch.uint8)
    w8[:, 5] = codes
    w8[:, 300] = codes.flip(0)
    s = torch.ones(256)
    for col, exp in ((5, torch.where(finite, want, torch.zeros(()))), (300, torch.where(finite, want, torch.zeros(())).flip(0))):
        x = torch.zeros(1, K)
        x[0, col] = 1.0
        got = ops.linear_skinny_fp8w(x.to(DEV), w8.to(DEV), s.to(DEV)).cpu()[0]
        assert torch.equal(got, exp), (col, (got != exp).nonzero()[:5])


@pytest.mark.parametrize("N,K", [(3072, 1024), (4096, 1024), (1024, 4096), (1025, 1024), (4608, 1536), (1536, 6144), (1536, 1536)])
def test_gemv_fp8w_bit_identical_to_bf16_on_dequantised_weights(N, K):
    q, s, deq = _quant(N, K, 1)
    wb = deq.to(torch.bfloat16)
    assert torch.equal(wb.float(), deq)
    x = _rand(1, K, seed=2)
    bias = _rand(N, seed=3) * 0.1
    g = _rand(K, seed=4) * 0.2 + 1.0
    b = _rand(K, seed=5) * 0.1
    r0 = _rand(1, N, seed=6)
    ref = x.double() @ deq.double().t() + bias.double()
    got = ops.linear_skinny_fp8w(x, q, s, bias, 2, resid=r0.clone())
    assert torch.equal(got, ops.linear_skinny(x, wb, bias, 2, resid=r0.clone()))
    assert (got.double() - (ref + r0.double())).abs().max().item() < 2e-4 * ref.abs().max().item()
    if K <= 2048:  # the K = d family carries every prologue / epilogue; K = 4d exists as linear2 (+ residual) only, like the bf16 GEMV
        assert torch.equal(ops.linear_skinny_fp8w(x, q, s, bias, 0), ops.linear_skinny(x, wb, bias, 0))
        assert torch.equal(ops.linear_skinny_fp8w(x, q, s, bias, 1), ops.linear_skinny(x, wb, bias, 1))
        assert torch.equal(ops.linear_skinny_fp8w(x, q, s, None, 0, gamma=g, beta=b), ops.linear_skinny(x, wb, None, 0, gamma=g, beta=b))


@pytest.mark.parametrize("M", [2, 8, 33, 64])
@pytest.mark.parametrize("N,K", [(3072, 1024), (1024, 4096), (1025, 1024), (4608, 1536), (1536, 6144)])
@pytest.mark.parametrize("epi", [ops.EPI_STORE, ops.EPI_RELU, ops.EPI_RESID, ops.EPI_F32])
def test_gemm_skinny_fp8w(M, N, K, epi):
    q, s, deq = _quant(N, K, 11)
    a = _rand(M, K, seed=12).to(torch.bfloat16)
    bias = _rand(N, seed=13) * 0.1
    r0 = _rand(M, N, seed=14)
    ref = a.double() @ deq.double().t() + bias.double()
    if epi == ops.EPI_RELU:
        ref = ref.clamp_min(0)
    if epi == ops.EPI_RESID:
        ref = ref + r0.double()
    kw = dict(resid=r0.clone()) if epi == ops.EPI_RESID else {}
    got = ops.linear_fp8w(a, q, s, bias, epi, **kw)
    same = ops.linear(a, deq.to(torch.bfloat16), bias, epi, **(dict(resid=r0.clone()) if epi == ops.EPI_RESID else {}))
    scale = max(1.0, ref.abs().max().item())
    tol = 0.01 * scale if epi in (ops.EPI_STORE, ops.EPI_RELU) else 3e-6 * math.sqrt(K) * scale
    assert (got.double() - ref).abs().max().item() < tol
    # the bf16 kernel on bf16(W'): same products, but the fp8 kernel assigns k to MFMA lanes differently (one 16-byte
    # vector = 16 consecutive k per lane), so only the fp32 summation order differs
    assert (got.double() - same.double()).abs().max().item() < tol


def _teacher_forced_logits(m, x, y, S, P, forced):
    eng = m.engine_for(x.shape[0], S, P)
    eng.set_option("trace_ar_logits", 1)
    eng.set_option("trace_nar_logits", 1)
    B = x.shape[0]
    eng.prefill(x.to(DEV), [S] * B, y.to(DEV), [P] * B)
    _, gl = eng.generate(top_k=1, forced=forced.to(DEV), forced_lens=[forced.shape[1]] * B)
    lg = eng.fetch_ar_logits().clone()
    codes = eng.nar(None).cpu()
    return lg, codes, [eng.fetch_nar_logits(i).clone() for i in range(7)]


@pytest.mark.parametrize("d,h,L", [(256, 4, 3), (1536, 16, 2)])
def test_engine_fp8w_equals_bf16_engine_on_dequantised_weights(d, h, L):
    """(b) of the module docstring; d = 1536, h = 16 (dh = 96) is the layer shape of BASELINE.json configs[4]."""
    # untied predict layers: a tied model cannot hold W' for nar_predict_layers[j] and the fp32 original for
    # nar_audio_embeddings[j + 2] in one Parameter, which is what loading fp8w_state_dict into the bf16 model needs
    cfg = vo.OracleConfig(d_model=d, nhead=h, num_layers=L, prefix_mode=1, share_embedding=False)
    sd = vo.make_state_dict(cfg, 21)
    sdq = vo.fp8w_state_dict(sd)
    S, P, G = 9, 21, 20
    x, xl, y = vo.make_inputs(S, P, seed=31)
    forced = torch.randint(0, 1024, (1, G), generator=torch.Generator().manual_seed(7))
    m8 = build_model(cfg, sd, "fp8w")
    lg8, codes8, nar8 = _teacher_forced_logits(m8, x, y, S, P, forced)
    del m8
    mb = build_model(cfg, sdq, "bf16")
    lgb, codesb, narb = _teacher_forced_logits(mb, x, y, S, P, forced)
    assert torch.equal(lg8, lgb), (lg8 - lgb).abs().max()
    assert torch.equal(codes8, codesb)
    for a, b in zip(nar8, narb):
        assert torch.equal(a, b)
    # free-running greedy decode: identical tokens
    out8 = build_model(cfg, sd, "fp8w").inference_batch(x.to(DEV), xl, y.to(DEV), [P], None, top_k=1, max_new=G)[0].cpu()
    outb = mb.inference_batch(x.to(DEV), xl, y.to(DEV), [P], None, top_k=1, max_new=G)[0].cpu()
    assert torch.equal(out8, outb)


@pytest.mark.parametrize("B", [3, 16])
def test_engine_fp8w_batch_path_matches_bf16_on_dequantised_weights(B):
    cfg = vo.OracleConfig(d_model=256, nhead=4, num_layers=2, prefix_mode=1, share_embedding=False)
    sd = vo.make_state_dict(cfg, 22)
    sdq = vo.fp8w_state_dict(sd)
    S, P, G = 6, 17, 12
    X = torch.stack([vo.make_inputs(S, P, seed=40 + b)[0][0] for b in range(B)])
    Y = torch.stack([vo.make_inputs(S, P, seed=40 + b)[2][0] for b in range(B)])
    forced = torch.randint(0, 1024, (B, G), generator=torch.Generator().manual_seed(8))
    lg8, codes8, _ = _teacher_forced_logits(build_model(cfg, sd, "fp8w", max_batch=B), X, Y, S, P, forced)
    lgb, codesb, _ = _teacher_forced_logits(build_model(cfg, sdq, "bf16", max_batch=B), X, Y, S, P, forced)
    # gemm_skinny: same products, different fp32 summation order inside the MFMA (k-to-lane assignment)
    sigma = lgb.std().item()
    # (a rounding-order difference flips a bf16 rounding of an activation now and then: 1 % of sigma, a fifth of the bf16 bar)
    assert (lg8 - lgb).abs().max().item() <= 1e-2 * sigma, ((lg8 - lgb).abs().max().item(), sigma)
    assert (lg8 - lgb).abs().mean().item() <= 1e-4 * sigma
    assert (codes8 == codesb).float().mean().item() > 0.97


def test_engine_fp8w_against_oracle_on_dequantised_weights():
    """(c): FP8W engine vs the fp32 CPU oracle run on W', teacher-forced on the oracle's tokens: bf16-mode bars."""
    cfg = vo.OracleConfig(d_model=256, nhead=4, num_layers=3, prefix_mode=1)
    sd = vo.make_state_dict(cfg, 23)
    sdq = vo.fp8w_state_dict(sd)
    S, P, G = 8, 20, 40
    x, xl, y = vo.make_inputs(S, P, seed=33)
    trace = {}
    want = vo.inference(sdq, cfg, x, xl, y, None, top_k=1, kv_cache=True, max_new=G, trace=trace)
    ref_tokens = want[0, :, 0]
    ref_logits = torch.stack(trace["ar_logits"])[: ref_tokens.numel()]
    m = build_model(cfg, sd, "fp8w")
    lg, codes, _ = _teacher_forced_logits(m, x, y, S, P, ref_tokens[None])
    mine = lg[: ref_tokens.numel(), 0]
    sigma = ref_logits.std().item()
    d = (mine - ref_logits).abs()
    assert d.max().item() <= 0.05 * sigma and d.mean().item() <= 0.01 * sigma, (d.max().item(), d.mean().item(), sigma)
    assert (codes[0, :, 1:] == want[0, :, 1:]).float().mean().item() > 0.9  # NAR argmax agreement on the same AR tokens
    # and the quantisation itself is a real change of the model: the un-quantised oracle differs
    plain = vo.inference(sd, cfg, x, xl, y, None, top_k=1, kv_cache=True, max_new=G)
    print("fp8w vs unquantised fp32 oracle, token agreement:", (plain[:, : want.shape[1]] == want[:, : plain.shape[1]]).float().mean().item())


def test_fp8w_step_bytes_are_half_of_bf16():
    cfg = vo.OracleConfig(d_model=256, nhead=4, num_layers=2, prefix_mode=1)
    sd = vo.make_state_dict(cfg, 1)
    e8 = build_model(cfg, sd, "fp8w").engine_for(1, 4, 4)
    eb = build_model(cfg, sd, "bf16").engine_for(1, 4, 4)
    w8, wb = e8.ar_step_bytes(1, 0), eb.ar_step_bytes(1, 0)
    assert 0.5 < w8 / wb < 0.56
