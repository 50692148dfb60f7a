"""GPU parity tests proper: the HIP engine (through the C ABI) against the committed golden
vectors of the reference (tests/golden) and against the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): greedy token ids bit-identical in fp32 mode; logits within a
stated tolerance (fp32: 2e-3 * max(1, sigma_logit) absolute; bf16: max 5% of sigma_logit,
teacher-forced on the reference's own token history)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import valle_amd  # noqa: E402
from oracle import valle_oracle as vo  # noqa: E402
from tests.golden_util import assert_persistent_launch_ran, list_cases, load_case  # noqa: E402

DEV = "cuda:0"
SMALL = [c for c in list_cases() if not c.startswith(("c1_", "c2_", "opt_", "vallf_"))]  # opt_* / vallf_*: tests/test_options_gpu.py (block-module decode)


def build_model(cfg, sd, dtype="fp32", **kw):
    m = valle_amd.VALLE(
        cfg.d_model, cfg.nhead, cfg.num_layers, prefix_mode=cfg.prefix_mode, share_embedding=cfg.share_embedding,
        prepend_bos=cfg.prepend_bos, num_quantizers=cfg.num_quantizers, engine_dtype=dtype, **kw,
    )
    m.load_state_dict(sd, strict=True)
    return m.to(DEV).eval()


def run_case(m, case):
    x, xl, y = case["x"].to(DEV), case["x_lens"].to(DEV), case["y"].to(DEV)
    if case["mode"] == "continual":
        return m.continual(x, xl, y).cpu()
    en = case["enroll"].to(DEV) if case["enroll"] is not None else None
    return m.inference(x, xl, y, en, top_k=case["top_k"], temperature=1.0).cpu()


@pytest.mark.parametrize("name", SMALL)
def test_fp32_matches_reference_golden(name):
    case = load_case(name)
    m = build_model(case["cfg"], case["sd"], "fp32")
    z = case["z"]
    trace = case["mode"] != "continual"
    if trace:
        S, P = int(z["S"]), int(z["P"])
        eng = m.engine_for(1, S, P)
        eng.set_option("trace_ar_logits", 1)
        eng.set_option("trace_nar_logits", 1)
    codes = run_case(m, case)
    assert codes.dtype == torch.int64 and codes.shape == case["codes"].shape, (codes.shape, case["codes"].shape)
    assert torch.equal(codes, case["codes"]), f"token ids differ: {(codes != case['codes']).sum().item()} of {codes.numel()}"
    if trace and "ar_logits" in z.files:
        mine = m._engine.fetch_ar_logits()[:, 0][:: int(z["ar_stride"])].numpy()
        assert mine.shape == z["ar_logits"].shape
        np.testing.assert_allclose(mine, z["ar_logits"], rtol=0, atol=2e-3 * max(1.0, float(z["ar_logit_std"])))
    if trace and "nar_logits" in z.files:
        rows = z["nar_rows"]
        for i in range(z["nar_logits"].shape[0]):
            mine = m._engine.fetch_nar_logits(i)[rows].numpy()
            np.testing.assert_allclose(mine, z["nar_logits"][i], rtol=0, atol=2e-3 * max(1.0, float(z["nar_logit_std"][i])))


def test_fp32_c1_d256_full_length_golden():
    """BASELINE.json configs[0] at full size: S=47, P=225 -> G=753, token ids exact."""
    case = load_case("c1_d256_L6")
    m = build_model(case["cfg"], case["sd"], "fp32")
    codes = run_case(m, case)
    assert codes.shape == (1, 753, 8)
    assert torch.equal(codes, case["codes"]), f"{(codes != case['codes']).sum().item()} token ids differ"


def test_c2_architecture_fp32_exact_and_bf16_teacher_forced():
    """dim1024-L12-h16 (BASELINE.json configs[1] architecture), shortened lengths."""
    case = load_case("c2_d1024_L12_short")
    z = case["z"]
    m = build_model(case["cfg"], case["sd"], "fp32")
    codes = run_case(m, case)
    assert torch.equal(codes, case["codes"])
    del m
    # bf16: teacher-forced on the reference's token history
    mb = build_model(case["cfg"], case["sd"], "bf16")
    S, P = int(z["S"]), int(z["P"])
    eng = mb.engine_for(1, S, P)
    eng.set_option("trace_ar_logits", 1)
    ref_tokens = case["codes"][0, :, 0]
    eng.prefill(case["x"].to(DEV), [S], case["y"].to(DEV), [P])
    _, gl = eng.generate(top_k=1, forced=ref_tokens[None].to(DEV), forced_lens=[ref_tokens.numel()])
    assert gl == [ref_tokens.numel()]
    assert_persistent_launch_ran(eng)  # bf16 at the C2 shape: the bars below are about the persistent launch
    stride = int(z["ar_stride"])
    mine = eng.fetch_ar_logits()[:, 0]
    sigma = float(z["ar_logit_std"])
    diff = (mine[::stride].numpy() - z["ar_logits"])
    assert np.abs(diff).max() <= 0.05 * sigma, (np.abs(diff).max(), sigma)
    assert np.abs(diff).mean() <= 0.01 * sigma
    # token equality wherever the reference's top1-top2 margin exceeds 2 * tau
    sampled = eng.fetch_sampled()[0, : ref_tokens.numel()]
    margin = torch.from_numpy(z["ar_margin"][: ref_tokens.numel()])
    safe = margin > 2 * 0.05 * sigma
    assert torch.equal(sampled[safe], ref_tokens[safe])
    agree = (sampled == ref_tokens).float().mean().item()
    print(f"bf16 teacher-forced argmax agreement {agree:.4f}; max|dlogit| {np.abs(diff).max():.4f} (sigma {sigma:.3f})")
    assert agree > 0.95


@pytest.mark.parametrize("dtype", ["fp32"])
@pytest.mark.parametrize("B", [3, 8, 12])
def test_ragged_batch_equals_independent_oracle_calls(B, dtype):
    """Batch extension: B independent utterances == B oracle calls."""
    cfg = vo.OracleConfig(d_model=128, nhead=2, num_layers=2, prefix_mode=1)
    sd = vo.make_state_dict(cfg, 5)
    g = torch.Generator().manual_seed(77)
    S = torch.randint(3, 8, (B,), generator=g).tolist()
    P = torch.randint(4, 20, (B,), generator=g).tolist()
    xs, ys, want = [], [], []
    for b in range(B):
        x, xl, y = vo.make_inputs(S[b], P[b], seed=100 + b)
        xs.append(x[0]); ys.append(y[0])
        want.append(vo.inference(sd, cfg, x, xl, y, None, top_k=1, kv_cache=True)[0])
    X = torch.zeros(B, max(S), dtype=torch.int64)
    Y = torch.zeros(B, max(P), 8, dtype=torch.int64)
    for b in range(B):
        X[b, : S[b]] = xs[b]
        Y[b, : P[b]] = ys[b]
    m = build_model(cfg, sd, dtype)
    got = m.inference_batch(X.to(DEV), torch.tensor(S, dtype=torch.int32), Y.to(DEV), P, None, top_k=1)
    for b in range(B):
        assert got[b].shape == want[b].shape, (b, got[b].shape, want[b].shape)
        assert torch.equal(got[b].cpu(), want[b]), f"utterance {b} differs"


@pytest.mark.parametrize("B", [2, 5, 20, 64])
def test_bf16_batch_path_teacher_forced_against_batch1(B):
    """bf16 AR step of 2..64 utterances (LayerNorm kernel + gemm_skinny.hip MFMA path, K/V written by the
    QKV epilogue) against the batch-1 path (gemv1.hip) of the same engine: the batch is teacher-forced on each
    utterance's own batch-1 greedy tokens; per-step logits within 3 % of sigma, final codes (AR + NAR) equal
    wherever no argmax flips.  Ragged text / prompt lengths."""
    cfg = vo.OracleConfig(d_model=256, nhead=4, num_layers=3, prefix_mode=1)
    sd = vo.make_state_dict(cfg, 9)
    g = torch.Generator().manual_seed(5)
    S = torch.randint(3, 7, (B,), generator=g).tolist()
    P = torch.randint(4, 40, (B,), generator=g).tolist()
    m = build_model(cfg, sd, "bf16", max_batch=B)
    X = torch.zeros(B, max(S), dtype=torch.int64)
    Y = torch.zeros(B, max(P), 8, dtype=torch.int64)
    inputs = []
    for b in range(B):
        x, xl, y = vo.make_inputs(S[b], P[b], seed=300 + b)
        X[b, : S[b]] = x[0]; Y[b, : P[b]] = y[0]
        inputs.append((x, y))
    X, Y = X.to(DEV), Y.to(DEV)
    eng = m.engine_for(B, max(S), max(P))
    eng.set_option("trace_ar_logits", 1)
    eng.set_option("ignore_eos", 1)
    nsteps = 24
    # batch-1 runs: tokens + logits
    toks, lg1 = [], []
    for b in range(min(B, 6)):
        eng.prefill(X[b : b + 1, : S[b]].contiguous(), [S[b]], Y[b : b + 1, : P[b]].contiguous(), [P[b]])
        c0, gl = eng.generate(top_k=1, max_new=nsteps)
        toks.append(c0[0, : gl[0]].clone()); lg1.append(eng.fetch_ar_logits()[: gl[0], 0].clone())
    # the batch, forced on those tokens (utterances >= 6 repeat utterance b % 6's tokens: only their shapes matter)
    F = torch.zeros(B, nsteps, dtype=torch.int64)
    for b in range(B):
        F[b] = toks[b % len(toks)][:nsteps].cpu()
    eng.prefill(X, S, Y, P)
    _, gl = eng.generate(top_k=1, forced=F.to(DEV), forced_lens=[nsteps] * B)
    assert gl == [nsteps] * B
    lgB = eng.fetch_ar_logits()
    for b in range(len(toks)):
        sigma = lg1[b].std().item()
        d = (lgB[:nsteps, b] - lg1[b][:nsteps]).abs().max().item()
        assert d <= 0.03 * sigma, (b, d, sigma)
    codes = eng.nar(None)
    assert codes.shape[0] == B and torch.equal(codes[: len(toks), :nsteps, 0].cpu(), F[: len(toks)])
    if B <= 5:
        # since round 6 the batched step never splits the KV stream by itself (choose_nsplit); the split + combine launches stay
        # reachable through option "nsplit" and must give the same logits up to the merge's fp32 re-association
        for ns in (2, 4):
            eng.set_option("nsplit", ns)
            eng.prefill(X, S, Y, P)
            eng.generate(top_k=1, forced=F.to(DEV), forced_lens=[nsteps] * B)
            lgS = eng.fetch_ar_logits()
            assert (lgS[:nsteps] - lgB[:nsteps]).abs().max().item() <= 2e-2 * lgB[:nsteps].std().item(), ns
        eng.set_option("nsplit", 0)


def _bench_inputs(i, S=47, P=225):
    g = torch.Generator().manual_seed(1234 + i)
    x = torch.randint(3, 100, (S,), generator=g, dtype=torch.int64)
    x[0], x[-1] = 1, 2
    y = torch.randint(0, 1024, (P, 8), generator=g, dtype=torch.int64)
    return x, y


def test_c2_full_size_properties_bf16():
    """BASELINE.json configs[1] at FULL size (d1024-L12-h16 bf16, S=47, P=225 -> G=16*S+1=753): the oracle cannot
    run this in seconds, so size-independent properties: length rule, id ranges, run-to-run determinism,
    hipGraph == eager launches, and the first AR codes unchanged by the NAR phase."""
    torch.manual_seed(0)
    m = valle_amd.VALLE(1024, 16, 12, prefix_mode=1, engine_dtype="bf16").to(DEV).eval()
    x, y = _bench_inputs(0)
    X, XL, Y = x[None].to(DEV), torch.tensor([47], dtype=torch.int32, device=DEV), y[None].to(DEV)
    a = m.inference(X, XL, Y, None, top_k=1).cpu()
    assert_persistent_launch_ran(m.engine_for(1, 47, 225))
    assert a.shape == (1, 753, 8) and a.dtype == torch.int64            # valle.py:1047 cap: 16 * 47 + 1 frames
    assert int(a.min()) >= 0 and int(a[..., 1:].max()) < 1024 and int(a[..., 0].max()) <= 1024
    b = m.inference(X, XL, Y, None, top_k=1).cpu()
    assert torch.equal(a, b), "two identical greedy decodes differ"
    eng = m.engine_for(1, 47, 225)
    eng.set_option("steps_per_graph", 3)                                  # different graph partition of the same steps
    c = m.inference(X, XL, Y, None, top_k=1).cpu()
    assert torch.equal(a, c)
    sd = m.state_dict()
    m2 = valle_amd.VALLE(1024, 16, 12, prefix_mode=1, engine_dtype="bf16", use_graph=False)
    m2.load_state_dict(sd, strict=True)
    d = m2.to(DEV).eval().inference(X, XL, Y, None, top_k=1).cpu()
    assert torch.equal(a, d), "hipGraph replay and eager launches disagree"


def test_c3_batch64_full_architecture_batch_invariance():
    """BASELINE.json configs[2] shape (d1024-L12-h16 bf16, 64 utterances, KV cache 64 x 50 MB): utterance b is a copy
    of utterance b % 4, so the 16 copies of each must come out bit-identical (rows of a batch are independent in
    every kernel), ragged generation lengths are honoured, and ids stay in range."""
    torch.manual_seed(0)
    B = 64
    m = valle_amd.VALLE(1024, 16, 12, prefix_mode=1, engine_dtype="bf16", max_batch=B).to(DEV).eval()
    S, P = 12, 60
    X = torch.stack([_bench_inputs(b % 4, S, P)[0] for b in range(B)]).to(DEV)
    Y = torch.stack([_bench_inputs(b % 4, S, P)[1] for b in range(B)]).to(DEV)
    eng = m.engine_for(B, S, P)
    eng.set_option("ignore_eos", 1)
    out = m.inference_batch(X, torch.full((B,), S, dtype=torch.int32), Y, [P] * B, None, top_k=1, max_new=40)
    assert len(out) == B
    for b in range(B):
        assert out[b].shape == (40, 8)
        assert torch.equal(out[b], out[b % 4]), f"utterance {b} differs from its copy {b % 4}"
        assert int(out[b].min()) >= 0 and int(out[b].max()) <= 1024
    assert not torch.equal(out[0], out[1])


def test_graph_and_eager_paths_agree():
    cfg = vo.OracleConfig(d_model=128, nhead=2, num_layers=2, prefix_mode=1)
    sd = vo.make_state_dict(cfg, 2)
    x, xl, y = vo.make_inputs(5, 9)
    a = build_model(cfg, sd, "bf16", use_graph=True).inference(x.to(DEV), xl.to(DEV), y.to(DEV), None, top_k=1)
    b = build_model(cfg, sd, "bf16", use_graph=False).inference(x.to(DEV), xl.to(DEV), y.to(DEV), None, top_k=1)
    assert torch.equal(a, b)


def test_sampled_decode_is_seeded_and_inside_topk():
    cfg = vo.OracleConfig(d_model=128, nhead=2, num_layers=2, prefix_mode=1)
    sd = vo.make_state_dict(cfg, 4)
    x, xl, y = vo.make_inputs(4, 8)
    m = build_model(cfg, sd, "fp32")
    eng = m.engine_for(1, 4, 8)
    eng.set_option("trace_ar_logits", 1)
    r1 = m.inference(x.to(DEV), xl.to(DEV), y.to(DEV), None, top_k=5, temperature=0.8, seed=11).cpu()
    logits = eng.fetch_ar_logits()[:, 0]
    r2 = m.inference(x.to(DEV), xl.to(DEV), y.to(DEV), None, top_k=5, temperature=0.8, seed=11).cpu()
    r3 = m.inference(x.to(DEV), xl.to(DEV), y.to(DEV), None, top_k=5, temperature=0.8, seed=12).cpu()
    assert torch.equal(r1, r2) and not torch.equal(r1[..., 0], r3[..., 0])
    G = r1.shape[1]
    top5 = torch.topk(logits[:G], 5, dim=-1).indices
    assert all(int(r1[0, t, 0]) in top5[t].tolist() for t in range(G))
    # the oracle, teacher-forced on the engine's sampled history, reproduces the engine's logits
    tr = {}
    vo.inference(sd, cfg, x, xl, y, None, top_k=1, kv_cache=True, force_tokens=r1[0, :, 0], trace=tr)
    ref = torch.stack(tr["ar_logits"])[:G]
    assert (ref - logits[:G]).abs().max().item() < 2e-3
    # no top-k filter (the reference default top_k=-100): still valid codes
    r4 = m.inference(x.to(DEV), xl.to(DEV), y.to(DEV), None, top_k=-100, temperature=1.0, seed=3).cpu()
    assert r4.min() >= 0 and r4.max() < 1024


def test_stop_rule_max_new_and_syntax_error():
    cfg = vo.OracleConfig(d_model=64, nhead=4, num_layers=1, prefix_mode=1)
    sd = vo.make_state_dict(cfg, 3)
    x, xl, y = vo.make_inputs(3, 5)
    m = build_model(cfg, sd, "fp32")
    full = m.inference(x.to(DEV), xl.to(DEV), y.to(DEV), None, top_k=1).cpu()
    assert full.shape == (1, vo.expected_gen_len(3), 8)
    part = m.inference_batch(x.to(DEV), xl, y.to(DEV), [5], None, top_k=1, max_new=10)[0].cpu()
    assert part.shape[0] == 10 and torch.equal(part[:, 0], full[0, :10, 0])
    # EOS as arg-max at the very first step -> SyntaxError like valle.py:1049-1052
    sd2 = dict(sd)
    w = torch.zeros_like(sd["ar_predict_layer.weight"])
    w[1024] = 1.0
    sd2["ar_predict_layer.weight"] = w
    sd2["ar_decoder.norm.weight"] = torch.zeros(64)
    sd2["ar_decoder.norm.bias"] = torch.ones(64)
    m2 = build_model(cfg, sd2, "fp32")
    with pytest.raises(SyntaxError):
        m2.inference(x.to(DEV), xl.to(DEV), y.to(DEV), None, top_k=1)


def test_reference_assertions_and_unsupported_configs():
    cfg = vo.OracleConfig(d_model=64, nhead=4, num_layers=1, prefix_mode=1)
    m = build_model(cfg, vo.make_state_dict(cfg, 0), "fp32")
    x, xl, y = vo.make_inputs(3, 5)
    with pytest.raises(AssertionError):
        m.inference(x[0].to(DEV), xl.to(DEV), y.to(DEV), None)  # x.ndim != 2 (valle.py:986)
    with pytest.raises(AssertionError):
        m.inference(x.to(DEV), xl.to(DEV), torch.cat([y, y]).to(DEV), None)  # batch != 1 (valle.py:989)
    with pytest.raises(RuntimeError):
        valle_amd.VALLE(64, 4, 1, norm_first=False).engine_for(1, 4, 4)  # post-norm decodes through the block modules


@pytest.mark.parametrize("top_k,temperature", [(5, 0.8), (40, 1.3), (-100, 1.0)])
def test_sampling_distribution_matches_reference_definition(top_k, temperature):
    """the RNG streams differ from torch.multinomial's, the DISTRIBUTION must not.  4096 first-step
    samples (64 identical utterances x 64 seeds) against softmax(top_k_filter(logits / temperature)) as the reference's
    topk_sampling defines it (valle.py:1242-1302, restated in the oracle): every token's frequency within 5 sigma of its
    probability, nothing sampled outside the filter's support."""
    cfg = vo.OracleConfig(d_model=64, nhead=4, num_layers=1, prefix_mode=1)
    sd = vo.make_state_dict(cfg, 8)
    x, xl, y = vo.make_inputs(4, 6)
    B, R = 64, 64
    m = build_model(cfg, sd, "fp32", max_batch=B)
    eng = m.engine_for(B, 4, 6)
    eng.set_option("trace_ar_logits", 1)
    X, Y = x.repeat(B, 1).to(DEV), y.repeat(B, 1, 1).to(DEV)
    counts = torch.zeros(1025, dtype=torch.float64)
    for seed in range(R):
        eng.prefill(X, [4] * B, Y, [6] * B)
        eng.generate(top_k=top_k, temperature=temperature, seed=1000 + seed, max_new=1, allow_empty=True)
        first = eng.fetch_sampled()[:, 0]
        counts += torch.bincount(first, minlength=1025).double()
    logits = eng.fetch_ar_logits()[0, 0]  # the first step's logits (identical for every utterance)
    scaled = logits / temperature if temperature != 1.0 else logits.clone()
    p = torch.softmax(vo.top_k_filtering(scaled[None].clone(), top_k), dim=-1)[0].double()
    n = float(B * R)
    assert counts.sum().item() == n
    assert counts[p == 0].sum().item() == 0, "sampled a token the top-k filter removed"
    sigma = torch.sqrt(n * p * (1 - p)).clamp_min(1.0)
    z = ((counts - n * p).abs() / sigma).max().item()
    assert z < 5.0, z
    if top_k > 0:
        assert int((p > 0).sum()) == top_k


@pytest.mark.parametrize("dtype", ["bf16", "fp8w"])
@pytest.mark.parametrize("B,d,h,L", [(5, 256, 4, 3), (64, 1024, 16, 2), (33, 1536, 16, 2)])
def test_fused_layernorm_batch_step_matches_layernorm_kernels(B, d, h, L, dtype):
    """The batched AR step with LayerNorm folded into the GEMMs (gemm_skinny.hip: producers emit bf16(x * gamma) + per-16-column
    statistics, consumers apply rstd * (acc - mean * W gamma) + W beta + b) against the same engine with the LayerNorm
    kernels (option gs_fuse_ln = 0), teacher-forced on the un-fused run's tokens: logits within 2 % of sigma at every step,
    bit-identical run to run, and the NAR codes / first codebook unchanged."""
    cfg = vo.OracleConfig(d_model=d, nhead=h, num_layers=L, prefix_mode=1)
    sd = vo.make_state_dict(cfg, 21)
    g = torch.Generator().manual_seed(9)
    S = torch.randint(3, 9, (B,), generator=g).tolist()
    P = torch.randint(4, 30, (B,), generator=g).tolist()
    X = torch.zeros(B, max(S), dtype=torch.int64)
    Y = torch.zeros(B, max(P), 8, dtype=torch.int64)
    for b in range(B):
        x, _, y = vo.make_inputs(S[b], P[b], seed=900 + b)
        X[b, : S[b]] = x[0]; Y[b, : P[b]] = y[0]
    X, Y = X.to(DEV), Y.to(DEV)
    m = build_model(cfg, sd, dtype, max_batch=B)
    eng = m.engine_for(B, max(S), max(P))
    eng.set_option("trace_ar_logits", 1)
    eng.set_option("ignore_eos", 1)
    n = 12
    eng.set_option("gs_fuse_ln", 0)
    eng.prefill(X, S, Y, P)
    c0, gl = eng.generate(top_k=1, max_new=n)
    ref_tok, ref_lg = c0[:, :n].clone(), eng.fetch_ar_logits()[:n + 1].clone()
    runs = []
    eng.set_option("gs_fuse_ln", 1)
    for _ in range(2):
        eng.prefill(X, S, Y, P)
        eng.generate(top_k=1, forced=ref_tok, forced_lens=[n] * B)
        runs.append(eng.fetch_ar_logits()[:n + 1].clone())
    assert torch.equal(runs[0], runs[1]), "fused-LayerNorm step is not deterministic"
    sigma = ref_lg.std().item()
    err = (runs[0] - ref_lg).abs().max().item()
    assert err <= 0.02 * sigma, (err, sigma)  # two bf16 roundings of different quantities: each within ~1 % of exact
    assert (runs[0][0] - ref_lg[0]).abs().max().item() == 0.0  # step 0 = the prefill's logits: same kernels in both modes


@pytest.mark.parametrize("dtype", ["bf16", "fp8w"])
@pytest.mark.parametrize("B,d,h", [(64, 1024, 16), (40, 1024, 16), (9, 1024, 16), (33, 1536, 16), (24, 1536, 16), (12, 1536, 16), (17, 512, 8)])
def test_skinny_gemm_compile_time_layout_body_is_bit_identical(B, d, h, dtype):
    """gemm_skinny.hip's FAST bodies (fragment-major W and X, rounds of 4 chunks + a round of 2, every layout decision at compile
    time; option gs_fast, default 1) against the general bodies of the same kernels on the batched AR step: d = 1024 (4 chunks
    per wave: QKV / FFN1 / FFN2 and the M-split out-proj all qualify), d = 1536 (4 + 2 chunks, 96 statistics slots; M-split
    with 3 chunks; at <= 32 utterances TWO W fragments per workgroup, the one-fragment grids of 288 / 384 workgroups not fitting the
    chip) and d = 512 (one round of 2; the LayerNorm consumers stay on the general body).  Same loads, same MFMA
    order, same epilogue arithmetic (explicit fma's): the logits of every step must be bit-identical -- also with the split-K
    hand-off through granules (option gs_gran) instead of the ticket, whose finisher must never time out."""
    L = 2
    cfg = vo.OracleConfig(d_model=d, nhead=h, num_layers=L, prefix_mode=1)
    sd = vo.make_state_dict(cfg, 22)
    g = torch.Generator().manual_seed(10)
    S = torch.randint(3, 9, (B,), generator=g).tolist()
    P = torch.randint(4, 30, (B,), generator=g).tolist()
    X = torch.zeros(B, max(S), dtype=torch.int64)
    Y = torch.zeros(B, max(P), 8, dtype=torch.int64)
    for b in range(B):
        x, _, y = vo.make_inputs(S[b], P[b], seed=950 + b)
        X[b, : S[b]] = x[0]; Y[b, : P[b]] = y[0]
    X, Y = X.to(DEV), Y.to(DEV)
    m = build_model(cfg, sd, dtype, max_batch=B)
    eng = m.engine_for(B, max(S), max(P))
    eng.set_option("trace_ar_logits", 1)
    eng.set_option("ignore_eos", 1)
    n = 10
    out = {}
    try:
        # gs_gran: the split-K hand-off (FFN2) through {tag, value} granules instead of the ticket -- same slice order, same sums
        for fast, gran in ((0, 0), (1, 0), (1, 1), (0, 1)):
            eng.set_option("gs_fast", fast)
            eng.set_option("gs_gran", gran)
            for _ in range(2):  # twice: the second decode starts on the granules the first one left behind
                eng.prefill(X, S, Y, P)
                c, _ = eng.generate(top_k=1, max_new=n)
            out[fast, gran] = (c[:, :n].clone(), eng.fetch_ar_logits()[:n + 1].clone())
    finally:
        eng.set_option("gs_fast", 1)
        eng.set_option("gs_gran", 0)
    for k in ((1, 0), (1, 1), (0, 1)):
        assert torch.equal(out[0, 0][0], out[k][0]), k
        assert torch.equal(out[0, 0][1], out[k][1]), (k, (out[0, 0][1] - out[k][1]).abs().max().item())
    assert torch.isfinite(out[1, 1][1]).all()
    fails = C.c_uint(123)
    assert eng.lib.vle_debug_fetch(eng.h, b"gs_gran_fail", C.byref(fails), 4) == 4 and fails.value == 0


@pytest.mark.parametrize("dtype", ["bf16", "fp8w"])
@pytest.mark.parametrize("fuse_ln", [1, 0])
@pytest.mark.parametrize("B,d,h,L", [(5, 256, 4, 3), (64, 1024, 16, 2), (33, 1024, 8, 2)])
def test_attention_with_fused_out_proj_matches_separate_launches(B, d, h, L, dtype, fuse_ln):
    """Option attn_oproj (a measured dead end kept for A/B -- the step gets SLOWER): the batched step's decode
    attention launch also does the layer's out-proj + residual (+ the LayerNorm
    producer): per (utterance, head) partial products with that head's slice of W_o, summed in head order by the utterance's last
    block (decode_attn.hip AttnOproj).  Against the same engine with the separate out-proj GEMM, teacher-forced on its tokens:
    the same bf16 products in another summation order -- logits within 1 % of sigma at every step, bit-identical run to run
    (ragged batch: arrival order of the heads varies), head sizes 64 and 128."""
    cfg = vo.OracleConfig(d_model=d, nhead=h, num_layers=L, prefix_mode=1)
    sd = vo.make_state_dict(cfg, 23)
    g = torch.Generator().manual_seed(10)
    S = torch.randint(3, 9, (B,), generator=g).tolist()
    P = torch.randint(4, 40, (B,), generator=g).tolist()
    X = torch.zeros(B, max(S), dtype=torch.int64)
    Y = torch.zeros(B, max(P), 8, dtype=torch.int64)
    for b in range(B):
        x, _, y = vo.make_inputs(S[b], P[b], seed=700 + b)
        X[b, : S[b]] = x[0]; Y[b, : P[b]] = y[0]
    X, Y = X.to(DEV), Y.to(DEV)
    m = build_model(cfg, sd, dtype, max_batch=B)
    eng = m.engine_for(B, max(S), max(P))
    eng.set_option("trace_ar_logits", 1)
    eng.set_option("ignore_eos", 1)
    eng.set_option("gs_fuse_ln", fuse_ln)
    n = 12
    eng.set_option("attn_oproj", 0)
    eng.prefill(X, S, Y, P)
    c0, gl = eng.generate(top_k=1, max_new=n)
    ref_tok, ref_lg = c0[:, :n].clone(), eng.fetch_ar_logits()[:n + 1].clone()
    runs = []
    eng.set_option("attn_oproj", 1)
    for _ in range(3):
        eng.prefill(X, S, Y, P)
        eng.generate(top_k=1, forced=ref_tok, forced_lens=[n] * B)
        runs.append(eng.fetch_ar_logits()[:n + 1].clone())
    eng.set_option("attn_oproj", 0)
    assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2]), "fused attention + out-proj is not deterministic"
    sigma = ref_lg.std().item()
    err = (runs[0] - ref_lg).abs().max().item()
    assert err <= 0.01 * sigma, (err, sigma)  # measured 0.1-0.5 %
    assert (runs[0][0] - ref_lg[0]).abs().max().item() == 0.0  # step 0 = the prefill's logits: same kernels in both modes


def test_engine_grows_capacities_without_reloading_weights():
    """engine_for() on a bigger request re-creates the buffers in place (vle_reserve) -- same engine object, same weights on the
    device -- and the decodes before / after equal the oracle; a batch-1 engine grows into the batched (fused-LayerNorm) path."""
    cfg = vo.OracleConfig(d_model=256, nhead=4, num_layers=2, prefix_mode=1)
    sd = vo.make_state_dict(cfg, 6)
    m = build_model(cfg, sd, "fp32")
    x, xl, y = vo.make_inputs(4, 6)
    a = m.inference(x.to(DEV), xl.to(DEV), y.to(DEV), None, top_k=1).cpu()
    eng0 = m._engine
    assert torch.equal(a, vo.inference(sd, cfg, x, xl, y, None, top_k=1, kv_cache=True))
    x2, xl2, y2 = vo.make_inputs(9, 30, seed=77)   # longer text (position table grows), longer prompt
    b = m.inference(x2.to(DEV), xl2.to(DEV), y2.to(DEV), None, top_k=1).cpu()
    assert m._engine is eng0 and eng0.cfg.max_text >= 9 and eng0.cfg.max_prompt >= 30
    assert torch.equal(b, vo.inference(sd, cfg, x2, xl2, y2, None, top_k=1, kv_cache=True))
    X = torch.zeros(3, 9, dtype=torch.int64); Y = torch.zeros(3, 30, 8, dtype=torch.int64)
    S, P, want = [4, 9, 6], [6, 30, 11], []
    for i, (s_, p_) in enumerate(zip(S, P)):
        xi, xli, yi = vo.make_inputs(s_, p_, seed=200 + i)
        X[i, :s_], Y[i, :p_] = xi[0], yi[0]
        want.append(vo.inference(sd, cfg, xi, xli, yi, None, top_k=1, kv_cache=True)[0])
    got = m.inference_batch(X.to(DEV), torch.tensor(S, dtype=torch.int32), Y.to(DEV), P, None, top_k=1)  # batch 1 -> 3
    assert m._engine is eng0 and eng0.cfg.max_batch >= 3
    for i in range(3):
        assert torch.equal(got[i].cpu(), want[i])
    assert torch.equal(m.inference(x.to(DEV), xl.to(DEV), y.to(DEV), None, top_k=1).cpu(), a)  # and the first request again
    # bf16: batch-1 engine grows into the batched step (weight packs + fused LayerNorm made at vle_reserve)
    mb = build_model(cfg, sd, "bf16")
    mb.inference(x.to(DEV), xl.to(DEV), y.to(DEV), None, top_k=1)
    e1 = mb._engine
    ob = mb.inference_batch(X.to(DEV), torch.tensor(S, dtype=torch.int32), Y.to(DEV), P, None, top_k=1)
    assert mb._engine is e1
    mb2 = build_model(cfg, sd, "bf16", max_batch=3, max_text=9, max_prompt=30)
    ref = mb2.inference_batch(X.to(DEV), torch.tensor(S, dtype=torch.int32), Y.to(DEV), P, None, top_k=1)
    for i in range(3):
        assert torch.equal(ob[i], ref[i]), "a grown engine and a fresh one of the same capacity disagree"


def test_failed_reserve_marks_the_engine_unusable_and_the_model_recovers():
    """vle_reserve frees the capacity-dependent buffers before it re-allocates; when the re-allocation fails (here: a
    KV cache of > 1 TB, refused by the very first hipMalloc -- nothing is consumed) the engine must not keep pointers into freed
    memory: every entry point answers VLE_ESTATE, and VALLE.engine_for drops the handle so the next request builds a new engine."""
    cfg = vo.OracleConfig(d_model=256, nhead=4, num_layers=2, prefix_mode=1)
    sd = vo.make_state_dict(cfg, 6)
    m = build_model(cfg, sd, "fp32")
    x, xl, y = vo.make_inputs(4, 6)
    want = vo.inference(sd, cfg, x, xl, y, None, top_k=1, kv_cache=True)
    assert torch.equal(m.inference(x.to(DEV), xl.to(DEV), y.to(DEV), None, top_k=1).cpu(), want)
    eng = m._engine
    with pytest.raises(RuntimeError):
        eng.reserve(2_000_000, 64, 64)
    with pytest.raises(RuntimeError, match="unusable|not finalized"):
        eng.prefill(x.to(DEV), [4], y.to(DEV), [6])
    with pytest.raises(RuntimeError):
        eng.reserve(1, 4, 6)
    # through the model: the failed growth invalidates the engine, the next call rebuilds it
    m2 = build_model(cfg, sd, "fp32")
    m2.inference(x.to(DEV), xl.to(DEV), y.to(DEV), None, top_k=1)
    with pytest.raises(RuntimeError):
        m2.engine_for(2_000_000, 64, 64)
    assert m2._engine is None
    assert torch.equal(m2.inference(x.to(DEV), xl.to(DEV), y.to(DEV), None, top_k=1).cpu(), want)


def test_nar_force_is_one_shot_on_every_exit_path():
    """a vle_nar_force pointer must not survive a failed vle_nar_decode (it is caller-owned memory), and its stride
    is validated against the generated lengths."""
    cfg = vo.OracleConfig(d_model=256, nhead=4, num_layers=2, prefix_mode=1)
    sd = vo.make_state_dict(cfg, 6)
    m = build_model(cfg, sd, "fp32")
    x, xl, y = vo.make_inputs(4, 6)
    want = vo.inference(sd, cfg, x, xl, y, None, top_k=1, kv_cache=True)
    eng = m.engine_for(1, 4, 6)
    eng.prefill(x.to(DEV), [4], y.to(DEV), [6])
    import ctypes as C

    first, lens = eng.generate(top_k=1)
    G = int(lens[0])
    assert G >= 2
    short = torch.zeros(1, G - 1, 8, dtype=torch.int64, device=DEV)  # stride smaller than G: rejected by the C ABI
    assert eng.lib.vle_nar_force(eng.h, C.c_void_p(short.data_ptr()), G - 1) == 0
    with pytest.raises(RuntimeError, match="f_stride"):
        eng.nar()
    del short
    codes = eng.nar()  # the rejected pointer was forgotten: an ordinary arg-max decode
    assert torch.equal(codes[0][:G].cpu(), want[0])


def test_fused_qkv_attention_launch_and_its_q_handoff_agree_with_the_separate_launches():
    """Batch-1 AR step, C2 architecture (d1024-L12-h16), bf16 and fp32: the fused LN1 + QKV + attention launch with the in-launch
    hand-off of q (default), the same launch recomputing q per KV split (qa_handoff = 0), 4 / 8 / 16 splits, and the two separate
    launches (qkv_attn = 0) produce the same teacher-forced logits up to fp32 summation order; no workgroup of the hand-off ever
    had to give up (qa_spin_fail == 0); a second utterance after the first (epochs restart at the prefill) is still right."""
    import ctypes as C

    case = load_case("c2_d1024_L12_short")
    z = case["z"]
    S, P = int(z["S"]), int(z["P"])
    forced = case["codes"][:, :, 0].contiguous()
    G = forced.shape[1]
    for dtype, tol in (("bf16", 2e-3), ("fp32", 2e-4)):
        m = build_model(case["cfg"], case["sd"], dtype)
        eng = m.engine_for(1, S, P)
        eng.set_option("trace_ar_logits", 1)

        def run(**opts):
            for k, v in dict(qkv_attn=1, qa_handoff=1, qa_nsplit=8).items():
                eng.set_option(k, v)
            for k, v in opts.items():
                eng.set_option(k, v)
            eng.prefill(case["x"].to(DEV), [S], case["y"].to(DEV), [P])
            eng.generate(top_k=1, forced=forced.to(DEV), forced_lens=[G])
            return eng.fetch_ar_logits()[:, 0].clone()

        base = run()
        sigma = base.std().item()
        assert torch.equal(run(), base)  # a second utterance on the same engine: the granules' epochs restart
        for opts in (dict(qa_handoff=0), dict(qa_nsplit=4), dict(qa_nsplit=16), dict(qa_handoff=0, qa_nsplit=4), dict(qkv_attn=0)):
            other = run(**opts)
            assert (other - base).abs().max().item() <= tol * sigma, (dtype, opts, (other - base).abs().max().item(), sigma)
        fails = C.c_uint(0)
        assert eng.lib.vle_debug_fetch(eng.h, b"qa_spin_fail", C.byref(fails), 4) == 4 and fails.value == 0


@pytest.mark.parametrize("d,h,dtype", [(256, 2, "fp32"), (256, 2, "bf16"), (512, 4, "fp32"), (2048, 16, "bf16")])
def test_batch1_step_where_the_fused_launch_lacks_its_out_proj_gemv(d, h, dtype):
    """Shapes the fused QKV + attention launch covers but whose follow-up out-proj GEMV (the PRO_ATTN_SELF prologue) is not
    instantiated (fp32 d256-h2: 128-wide heads over one-element thread slices) must take the two separate launches, and every
    accepted qa_nsplit must decode: fp32 token ids equal to the oracle's, bf16 teacher-forced within 5 % of sigma."""
    cfg = vo.OracleConfig(d_model=d, nhead=h, num_layers=2, prefix_mode=1)
    sd = vo.make_state_dict(cfg, 31)
    x, xl, y = vo.make_inputs(6, 14, seed=5)
    tr = {}
    want = vo.inference(sd, cfg, x, xl, y, None, top_k=1, kv_cache=True, trace=tr)
    ref = torch.stack(tr["ar_logits"])
    m = build_model(cfg, sd, dtype)
    eng = m.engine_for(1, 6, 14)
    eng.set_option("trace_ar_logits", 1)
    for ns in (4, 8, 16):
        eng.set_option("qa_nsplit", ns)
        eng.prefill(x.to(DEV), [6], y.to(DEV), [14])
        G = want.shape[1]
        if dtype == "fp32":
            codes0, gl = eng.generate(top_k=1)
            assert gl == [G] and torch.equal(codes0[0, :G].cpu(), want[0, :, 0]), (d, h, ns)
        else:
            _, gl = eng.generate(top_k=1, forced=want[:, :, 0].to(DEV), forced_lens=[G])
            assert gl == [G]
            mine = eng.fetch_ar_logits()[:, 0]
            err = (mine - ref).abs().max().item()
            assert err <= 0.05 * ref.std().item(), (d, h, ns, err)


@pytest.mark.parametrize("dtype", ["bf16", "fp8w"])
def test_layernorm_folded_into_the_packed_row_gemms_matches_the_layernorm_launches(dtype):
    """Option "ln_fold" (default 1): the prefill and the 7 NAR passes run 5 launches per layer -- the LayerNorms ride on the residual
    GEMMs' epilogues (kernels.h GemmLn).  Against the same engine with the LayerNorm launches (ln_fold = 0), teacher-forced on one
    history: the prefill's logits and every NAR stage's logits within 2 % of their spread (two bf16 roundings of different
    quantities, each within ~1 % of exact), codes equal wherever the margin allows; and deterministic run to run."""
    cfg = vo.OracleConfig(d_model=256, nhead=4, num_layers=3, prefix_mode=1)
    sd = vo.make_state_dict(cfg, 21)
    S, P, G = 24, 110, 60  # 134 prefill rows, 194 NAR rows: the folded path needs >= 128 packed rows
    x, xl, y = vo.make_inputs(S, P, seed=5)
    m = build_model(cfg, sd, dtype)
    eng = m.engine_for(1, S, P)
    for k in ("trace_ar_logits", "trace_nar_logits", "ignore_eos"):
        eng.set_option(k, 1)
    X, Y = x.to(DEV), y.to(DEV)

    def run(fold, forced=None, nar_forced=None):
        eng.set_option("ln_fold", fold)
        eng.prefill(X, [S], Y, [P])
        kw = dict(forced=forced, forced_lens=[G]) if forced is not None else dict(max_new=G)
        c0, gl = eng.generate(top_k=1, **kw)
        ar = eng.fetch_ar_logits()[:, 0].clone()
        codes = eng.nar(None, forced=nar_forced).clone()
        return c0[:, : gl[0]].clone(), ar, codes, [eng.fetch_nar_logits(i).clone() for i in range(7)]

    tok, ar0, codes0, nar0 = run(0)
    _, ar1, codes1, nar1 = run(1, forced=tok, nar_forced=codes0)
    _, ar2, codes2, nar2 = run(1, forced=tok, nar_forced=codes0)
    assert torch.equal(ar1, ar2) and torch.equal(codes1, codes2) and all(torch.equal(a, b) for a, b in zip(nar1, nar2))
    sig = ar0.std().item()
    assert (ar1[0] - ar0[0]).abs().max().item() <= 0.02 * sig, "prefill logits"
    assert (ar1 - ar0).abs().max().item() <= 0.03 * sig, "AR steps on the folded prefill's K/V cache"
    for i in range(7):
        s_i = nar0[i].std().item()
        assert (nar1[i] - nar0[i]).abs().max().item() <= 0.02 * s_i, (i, (nar1[i] - nar0[i]).abs().max().item(), s_i)
    assert (codes1 == codes0).float().mean().item() > 0.97
    eng.set_option("ln_fold", 1)
