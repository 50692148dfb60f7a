"""GPU: the constructor combinations outside the fused engine's shape -- post-norm layers (valle/modules/transformer.py:303-308),
prenets (valle/models/valle.py:99-126, 182-219), a NAR decoder of another width (nar_scale_factor, :83, :235, :241) -- and VALL-F
(cross-attention decoders, valle.py:50-710), decoded by the HIP block modules (valle_amd/model.py ``_inference_blocks``) against
the reference's outputs (tests/golden/opt_*.npz, vallf_*.npz, made by oracle/make_golden.py) and against the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import valle_amd  # noqa: E402
from valle_amd import modules as M  # noqa: E402
from valle_amd import ops  # noqa: E402
from oracle import valle_oracle as vo  # noqa: E402
from tests.golden_util import list_cases, load_case  # noqa: E402

DEV = "cuda:0"
OPT = [c for c in list_cases() if c.startswith(("opt_", "vallf_"))]


def build(cfg, sd, dtype="fp32"):
    m = (valle_amd.VALLF if cfg.model == "vallf" else valle_amd.VALLE)(cfg.d_model, cfg.nhead, cfg.num_layers, norm_first=cfg.norm_first, add_prenet=cfg.add_prenet,
                        prefix_mode=cfg.prefix_mode, share_embedding=cfg.share_embedding, nar_scale_factor=cfg.nar_scale_factor,
                        prepend_bos=cfg.prepend_bos, num_quantizers=cfg.num_quantizers, engine_dtype=dtype)
    m.load_state_dict(sd, strict=True)
    return m.to(DEV).eval()


def run(m, case, **kw):
    x, xl, y = case["x"].to(DEV), case["x_lens"].to(DEV), case["y"].to(DEV)
    if case["mode"] == "continual":
        return m.continual(x, xl, y).cpu()
    en = case["enroll"].to(DEV) if case["enroll"] is not None else None
    return m.inference(x, xl, y, en, top_k=case["top_k"], temperature=1.0, **kw).cpu()


def test_golden_cases_exist():
    assert len([c for c in OPT if c.startswith("opt_")]) >= 6 and len([c for c in OPT if c.startswith("vallf_")]) >= 3


@pytest.mark.parametrize("name", OPT)
def test_fp32_block_decode_matches_reference_golden(name):
    """Greedy token ids of the reference, bit for bit (fp32 kernels), for every option combination of the fixtures."""
    case = load_case(name)
    m = build(case["cfg"], case["sd"], "fp32")
    assert not m.fused
    codes = run(m, case)
    assert codes.dtype == torch.int64 and codes.shape == case["codes"].shape, (codes.shape, case["codes"].shape)
    bad = (codes != case["codes"])
    if bad.any():  # an arg-max flipped: only acceptable where the reference's own top-1 / top-2 margin is fp32 summation noise
        z = case["z"]
        first = int(bad.any(dim=-1)[0].nonzero()[0])
        margin = float(z["ar_margin"][first]) if "ar_margin" in z.files and bad[0, first, 0] else float(z["nar_margin"].min())
        assert margin < 2e-5, f"{int(bad.sum())} token ids differ, first at frame {first} (reference margin {margin:.2e})"


@pytest.mark.parametrize("name", ["opt_postnorm_prenet_pm1", "opt_scale2_pm1"])
def test_bf16_block_decode_follows_the_reference(name):
    """bf16 kernels, free running: the generated length is the reference's (cap rule) and the first-codebook stream agrees
    with the fp32 reference until a small-margin arg-max flips (random-init logits: sigma ~0.6, bf16 noise ~3 % sigma)."""
    case = load_case(name)
    m = build(case["cfg"], case["sd"], "bf16")
    codes = run(m, case)
    assert codes.shape == case["codes"].shape
    same = (codes[0, :, 0] == case["codes"][0, :, 0])
    first_diff = int((~same).nonzero()[0]) if not bool(same.all()) else codes.shape[1]
    assert first_diff >= 4, first_diff
    if first_diff < codes.shape[1]:
        assert float(case["z"]["ar_margin"][first_diff]) < 0.25 * float(case["z"]["ar_logit_std"]) + 0.05


@pytest.mark.parametrize("dtype,tol", [("fp32", 2e-5), ("bf16", 0.04)])
@pytest.mark.parametrize("adaptive", [False, True])
def test_post_norm_layer_matches_oracle(dtype, tol, adaptive):
    """TransformerEncoderLayer, post-norm branch, against the oracle's restatement of transformer.py:303-308."""
    d, h, T = 128, 4, 37
    cfg = vo.OracleConfig(d_model=d, nhead=h, num_layers=1, norm_first=False)
    sd = vo.make_state_dict(cfg, 5)
    prefix = "nar_decoder.layers.0" if adaptive else "ar_decoder.layers.0"
    layer = M.TransformerEncoderLayer(d, h, dim_feedforward=4 * d, batch_first=True, norm_first=False, adaptive_layer_norm=adaptive)
    layer.load_state_dict({k[len(prefix) + 1:]: v for k, v in sd.items() if k.startswith(prefix + ".")}, strict=True)
    layer = M.set_compute_dtype(layer.to(DEV).eval(), dtype)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(T, d, generator=g)
    stage = sd["nar_stage_embeddings.2.word_embeddings.weight"] if adaptive else None
    mask = vo.prefix_lm_mask(9, T - 9)
    want = vo.encoder_layer(sd, prefix, x, h, mask, stage, norm_first=False)
    src = (x[None].to(DEV), None if stage is None else stage.to(DEV))
    got = layer(src, src_mask=mask.to(DEV))[0][0].cpu()
    assert (got - want).abs().max().item() < tol * max(1.0, want.abs().max().item())


def test_prenets_match_oracle():
    """TextPrenet (3 x conv5 + BatchNorm(eval) + ReLU, Linear) and AudioPrenet as fp32 GEMMs against F.conv1d / F.batch_norm."""
    d = 64
    cfg = vo.OracleConfig(d_model=d, nhead=4, num_layers=1, add_prenet=True)
    sd = vo.make_state_dict(cfg, 9)
    tp, apn = M.TextPrenet(d), M.AudioPrenet(d)
    tp.load_state_dict({k[len("ar_text_prenet."):]: v for k, v in sd.items() if k.startswith("ar_text_prenet.")}, strict=True)
    apn.load_state_dict({k[len("ar_audio_prenet."):]: v for k, v in sd.items() if k.startswith("ar_audio_prenet.")}, strict=True)
    tp, apn = tp.to(DEV).eval(), apn.to(DEV).eval()
    g = torch.Generator().manual_seed(4)
    for T in (1, 3, 11, 200):
        x = torch.randn(T, d, generator=g)
        want = vo.text_prenet(sd, "ar_text_prenet", x)
        got = tp(x[None].to(DEV))[0].cpu()
        assert (got - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item()), T
        wa = vo.audio_prenet(sd, "ar_audio_prenet", x)
        ga = apn(x[None].to(DEV))[0].cpu()
        assert (ga - wa).abs().max().item() < 2e-5 * max(1.0, wa.abs().max().item()), T
    with pytest.raises(NotImplementedError):
        tp.train()(x[None].to(DEV))


@pytest.mark.parametrize("top_k,temperature", [(5, 0.8), (40, 1.3), (-100, 1.0), (1, 1.0)])
def test_topk_sample_operator_distribution(top_k, temperature):
    """vle_op_topk_sample: 8192 draws (rows x steps) of one logits row against softmax(top_k_filter(logits / temperature)) as the
    reference's topk_sampling defines it (valle.py:1242-1302); the arg-max output is torch.argmax's; draws are reproducible."""
    g = torch.Generator().manual_seed(11)
    row = torch.randn(1025, generator=g) * 1.5
    rows = 512
    logits = row[None].repeat(rows, 1).to(DEV)
    counts = torch.zeros(1025, dtype=torch.float64)
    for step in range(16):
        smp, am = ops.topk_sample(logits, top_k, temperature, seed=77, step=step)
        assert bool((am.cpu() == int(row.argmax())).all())
        counts += torch.bincount(smp.cpu(), minlength=1025).double()
    again, _ = ops.topk_sample(logits, top_k, temperature, seed=77, step=15)
    assert torch.equal(again, smp)
    scaled = row / temperature if temperature != 1.0 else row.clone()
    p = torch.softmax(vo.top_k_filtering(scaled[None].clone(), top_k), dim=-1)[0].double()
    n = float(rows * 16)
    assert counts[p == 0].sum().item() == 0, "sampled a token the top-k filter removed"
    sigma = torch.sqrt(n * p * (1 - p)).clamp_min(1.0)
    assert ((counts - n * p).abs() / sigma).max().item() < 5.0
    if top_k == 1:
        assert counts[int(row.argmax())] == n


def test_block_path_seed_matches_the_engine_stream_of_every_batch_position():
    """model._request_seed_base: the block path samples ONE row per call, so utterance b passes the seed whose
    request-0 stream is request b's stream of the call seed -- row b of a batched draw (what the fused engine / serving path
    draw for utterance b) must equal row 0 of the single-row draw with that seed, at every step."""
    from valle_amd.model import _request_seed_base

    g = torch.Generator().manual_seed(5)
    logits = (torch.randn(6, 1025, generator=g) * 2.0).to(DEV)
    for seed in (0, 77, 2**63 + 12345):
        for step in (0, 3, 400):
            batch, _ = ops.topk_sample(logits, -100, 1.0, seed=seed, step=step)
            for b in range(6):
                one, _ = ops.topk_sample(logits[b : b + 1], -100, 1.0, seed=_request_seed_base(seed, b), step=step)
                assert int(one[0]) == int(batch[b]), (seed, step, b)
    assert _request_seed_base(77, 0) == 77


def test_sampled_block_decode_is_reproducible_and_seeded():
    case = load_case("opt_prenorm_prenet_pm1")
    m = build(case["cfg"], case["sd"], "fp32")
    x, xl, y = case["x"].to(DEV), case["x_lens"].to(DEV), case["y"].to(DEV)
    a = m.inference(x, xl, y, None, top_k=20, temperature=0.9, seed=5).cpu()
    b = m.inference(x, xl, y, None, top_k=20, temperature=0.9, seed=5).cpu()
    c = m.inference(x, xl, y, None, top_k=20, temperature=0.9, seed=6).cpu()
    assert torch.equal(a, b)
    assert a.shape != c.shape or not torch.equal(a, c)
    torch.manual_seed(3)
    d1 = m.inference(x, xl, y, None, top_k=20, temperature=0.9).cpu()  # seed drawn from torch's global generator
    torch.manual_seed(3)
    d2 = m.inference(x, xl, y, None, top_k=20, temperature=0.9).cpu()
    assert torch.equal(d1, d2)


def test_forward_on_option_models_matches_oracle():
    """VALLE.forward (teacher-forced scoring) on a post-norm + prenet + half-width-NAR model against the oracle's forward."""
    cfg = vo.OracleConfig(d_model=64, nhead=4, num_layers=2, norm_first=False, add_prenet=True, nar_scale_factor=0.5, prefix_mode=1)
    sd = vo.make_state_dict(cfg, 2)
    m = build(cfg, sd, "fp32")
    x, xl, y = vo.make_inputs(6, 24)
    yl = torch.tensor([24], dtype=torch.int32)
    (_, _), loss, metrics = m(x.to(DEV), xl, y.to(DEV), yl, nar_stage=3, prefix_len=7)
    want, wm = vo.forward(sd, cfg, x, xl, y, yl, nar_stage=3, prefix_len=7)
    assert abs(float(loss) - float(want)) <= 2e-5 * abs(float(want))
    for k, v in wm.items():
        assert abs(float(metrics[k]) - v) <= 1e-4


def test_batch_api_on_option_models():
    """inference_batch on a non-fused model: utterance by utterance through the block modules, each equal to its own call."""
    case = load_case("opt_postnorm_prenet_pm1")
    m = build(case["cfg"], case["sd"], "fp32")
    x, xl, y = case["x"], case["x_lens"], case["y"]
    x2, xl2, y2 = vo.make_inputs(int(xl[0]) - 2, y.shape[1] - 3, seed=99)
    X = torch.zeros(2, x.shape[1], dtype=torch.int64)
    X[0], X[1, : x2.shape[1]] = x[0], x2[0]
    Y = torch.zeros(2, y.shape[1], 8, dtype=torch.int64)
    Y[0], Y[1, : y2.shape[1]] = y[0], y2[0]
    outs = m.inference_batch(X.to(DEV), torch.tensor([int(xl[0]), int(xl2[0])]), Y.to(DEV), [y.shape[1], y2.shape[1]], None, top_k=1)
    assert torch.equal(outs[0].cpu(), case["codes"][0])
    want1 = vo.inference(case["sd"], case["cfg"], x2, xl2, y2, None, top_k=1, kv_cache=True)
    assert torch.equal(outs[1].cpu(), want1[0])


@pytest.mark.parametrize("dtype,tol", [("fp32", 2e-5), ("bf16", 0.04)])
@pytest.mark.parametrize("norm_first", [True, False])
@pytest.mark.parametrize("adaptive", [False, True])
def test_decoder_layer_matches_oracle(dtype, tol, norm_first, adaptive):
    """TransformerDecoderLayer (self-attention + cross-attention over the memory + FFN, valle/modules/transformer.py:409-616) against
    the oracle's restatement, pre- and post-norm, plain and adaptive norms."""
    d, h, T, S = 128, 4, 29, 11
    cfg = vo.OracleConfig(d_model=d, nhead=h, num_layers=1, norm_first=norm_first, model="vallf")
    sd = vo.make_state_dict(cfg, 6)
    prefix = "nar_decoder.layers.0" if adaptive else "ar_decoder.layers.0"
    layer = M.TransformerDecoderLayer(d, h, dim_feedforward=4 * d, batch_first=True, norm_first=norm_first, adaptive_layer_norm=adaptive)
    layer.load_state_dict({k[len(prefix) + 1:]: v for k, v in sd.items() if k.startswith(prefix + ".")}, strict=True)
    layer = M.set_compute_dtype(layer.to(DEV).eval(), dtype)
    g = torch.Generator().manual_seed(8)
    x, mem = torch.randn(T, d, generator=g), torch.randn(S, d, generator=g)
    stage = sd["nar_stage_embeddings.1.word_embeddings.weight"] if adaptive else None
    mask = torch.triu(torch.ones(T, T, dtype=torch.bool), diagonal=1)
    want = vo.decoder_layer(sd, prefix, x, mem, h, mask, stage, norm_first)
    got = layer((x[None].to(DEV), None if stage is None else stage.to(DEV)), mem[None].to(DEV), tgt_mask=mask.to(DEV))[0][0].cpu()
    assert (got - want).abs().max().item() < tol * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("nhead,dh", [(4, 16), (2, 64), (3, 96), (2, 128)])
def test_cross_attention_operator(nhead, dh, dt):
    d, T, S = nhead * dh, 37, 23
    g = torch.Generator().manual_seed(12)
    q, kv = torch.randn(T, d, generator=g).to(dt), torch.randn(S, 2 * d, generator=g).to(dt)
    got = ops.cross_attention(q.to(DEV), kv.to(DEV), nhead).cpu().double()
    qd, k, v = q.double().view(T, nhead, dh), kv[:, :d].double().view(S, nhead, dh), kv[:, d:].double().view(S, nhead, dh)
    p = torch.softmax(torch.einsum("thd,shd->hts", qd, k) / dh ** 0.5, dim=-1)
    want = torch.einsum("hts,shd->thd", p, v).reshape(T, d)
    assert (got - want).abs().max().item() < (2e-6 if dt == torch.float32 else 0.02)


def test_vallf_surface():
    case = load_case("vallf_prenorm_pm1")
    m = build(case["cfg"], case["sd"], "fp32")
    assert isinstance(m, valle_amd.VALLF) and not m.fused
    with pytest.raises(NotImplementedError):
        m.continual(case["x"].to(DEV), case["x_lens"].to(DEV), case["y"].to(DEV))
    with pytest.raises(RuntimeError):
        m.engine_for(1, 4, 4)
