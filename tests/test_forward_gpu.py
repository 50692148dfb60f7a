"""GPU: the teacher-forced forward (VALLE.forward) on the HIP block modules + vle_op_cross_entropy
against the UNMODIFIED reference's losses (tests/golden/forward/*.npz) and against the oracle's metrics."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import valle_amd  # noqa: E402
from oracle import valle_oracle as vo  # noqa: E402
from oracle.make_golden_forward import CASES  # noqa: E402
from tests.test_forward_cpu import load_forward_case  # noqa: E402

DEV = "cuda:0"


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("dtype,rtol", [("fp32", 2e-5), ("bf16", 5e-3)])
def test_forward_loss_matches_reference(name, dtype, rtol):
    z, cfg, sd, x, xl, y, yl, kw = load_forward_case(name)
    cls = valle_amd.VALLF if cfg.model == "vallf" else valle_amd.VALLE  # VALLF.forward, valle.py:395-564
    m = cls(cfg.d_model, cfg.nhead, cfg.num_layers, norm_first=cfg.norm_first, add_prenet=cfg.add_prenet, prefix_mode=cfg.prefix_mode,
            prepend_bos=cfg.prepend_bos, engine_dtype=dtype)
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).eval()
    mkw = {k: v for k, v in kw.items() if k != "y_prompts"}  # the oracle's argument names, except prefix_mode 4's prompts:
    y_in, yl_in = y.to(DEV), yl.to(DEV)
    if cfg.prefix_mode == 4:                                  # they arrive as PromptedFeatures, like in the reference (valle.py:792-798)
        pr = kw["y_prompts"]
        y_in = valle_amd.PromptedFeatures(pr.to(DEV), y.to(DEV))
        yl_in = valle_amd.PromptedFeatures(torch.full((x.shape[0],), pr.shape[1], dtype=torch.int32), yl.to(DEV))
    (_, codes), loss, metrics = m(x.to(DEV), xl.to(DEV), y_in, yl_in, reduction="sum", **mkw)
    want = float(z["loss"])
    assert abs(float(loss) - want) <= rtol * abs(want), (float(loss), want)
    expect = y.clone()
    expect[torch.arange(y.shape[1])[None, :] >= yl.to(torch.int64)[:, None]] = 0  # padded frames are blanked (valle.py:811)
    if cfg.prefix_mode == 2 and "prompt_starts" in kw:        # the returned codes carry the blanked stretch of the target codebook (:370-373)
        P = min(225, int(0.25 * int(yl.min())))
        for n, st in enumerate(kw["prompt_starts"]):
            expect[n, st: st + P, kw["nar_stage"]] = 1024
    assert torch.equal(codes.cpu(), expect)
    _, ometrics = vo.forward(sd, cfg, x, xl, y, yl, **kw)
    for k, v in ometrics.items():
        tol = 1e-4 if dtype == "fp32" else 0.05 * float(x.shape[0] * y.shape[1])
        assert abs(float(metrics[k]) - v) <= tol, (k, float(metrics[k]), v)


def test_forward_default_draws_follow_the_reference_rng():
    """nar_stage / prefix_len left None: drawn like the reference (self.rng = random.Random(0) at construction, torch.randint)."""
    import random

    cfg = vo.OracleConfig(d_model=64, nhead=4, num_layers=1, prefix_mode=1)
    sd = vo.make_state_dict(cfg, 0)
    m = valle_amd.VALLE(64, 4, 1, prefix_mode=1, engine_dtype="fp32")
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).eval()
    x, xl, y = vo.make_inputs(5, 20)
    yl = torch.tensor([20], dtype=torch.int32)
    torch.manual_seed(7)
    (_, _), loss, _ = m(x.to(DEV), xl, y.to(DEV), yl)
    stage = random.Random(0).choices(list(range(1, 8)), weights=[1.0 / 7] * 7, k=1)[0]
    torch.manual_seed(7)
    plen = min(int(torch.randint(5, 10, size=()).item()), 225)
    want, _ = vo.forward(sd, cfg, x, xl, y, yl, train_stage=0, nar_stage=stage, prefix_len=plen)
    assert abs(float(loss) - float(want)) <= 2e-5 * abs(float(want))
    with pytest.raises(ValueError):
        m(x.to(DEV), torch.tensor([4], dtype=torch.int32), y.to(DEV), yl)  # lengths shorter than the tensors: not the collater's shapes
    mf = valle_amd.VALLF(64, 4, 1, prefix_mode=1, engine_dtype="fp32").to(DEV).eval()
    x2, y2 = torch.cat([x, x]), torch.cat([y, y])
    with pytest.raises(NotImplementedError):  # VALL-F scores unpadded batches only
        mf(x2.to(DEV), torch.tensor([5, 4], dtype=torch.int32), y2.to(DEV), torch.tensor([20, 20], dtype=torch.int32))


def test_forward_prefix_mode_2_draws_its_segment_starts_like_the_reference():
    """prompt_starts left None: one self.rng.randint(0, T - P) per utterance AFTER the nar_stage draw (valle.py:891-895, :368-369)."""
    import random

    cfg = vo.OracleConfig(d_model=64, nhead=4, num_layers=1, prefix_mode=2)
    sd = vo.make_state_dict(cfg, 0)
    m = valle_amd.VALLE(64, 4, 1, prefix_mode=2, engine_dtype="fp32")
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).eval()
    N, S, T = 2, 5, 24
    xs, ys = zip(*[(vo.make_inputs(S, T, 70 + b)[0][0], vo.make_inputs(S, T, 70 + b)[2][0]) for b in range(N)])
    x, y = torch.stack(xs), torch.stack(ys)
    xl, yl = torch.full((N,), S, dtype=torch.int32), torch.full((N,), T, dtype=torch.int32)
    (_, _), loss, _ = m(x.to(DEV), xl, y.to(DEV), yl)
    r = random.Random(0)
    stage = r.choices(list(range(1, 8)), weights=[1.0 / 7] * 7, k=1)[0]
    P = min(225, int(0.25 * T))
    starts = [r.randint(0, T - P) for _ in range(N)]
    want, _ = vo.forward(sd, cfg, x, xl, y, yl, train_stage=0, nar_stage=stage, prompt_starts=starts)
    assert abs(float(loss) - float(want)) <= 2e-5 * abs(float(want))
    with pytest.raises(ValueError):  # prefix_mode 4 without PromptedFeatures
        m4 = valle_amd.VALLE(64, 4, 1, prefix_mode=4, engine_dtype="fp32").to(DEV).eval()
        m4(x.to(DEV), xl, y.to(DEV), yl)
