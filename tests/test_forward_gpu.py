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
    (_, codes), loss, metrics = m(x.to(DEV), xl.to(DEV), y.to(DEV), yl.to(DEV), reduction="sum", **kw)
    want = float(z["loss"])
    assert abs(float(loss) - want) <= rtol * abs(want), (float(loss), want)
    assert torch.equal(codes.cpu(), y)
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
    with pytest.raises(NotImplementedError):
        m(x.to(DEV), torch.tensor([4], dtype=torch.int32), y.to(DEV), yl)  # padded text: not an unpadded batch
