"""GPU: the engine's stop on EOS against fixtures the UNMODIFIED reference produced (tests/golden/eos, oracle/make_eos_golden.py):
greedy decodes whose arg-max reaches EOS after 17 / 40 / 28 frames (prefix modes 1, 0 + BOS, 2 + enrolment) -- the reference stops
without appending EOS and runs its NAR stages on the shorter sequence (valle/models/valle.py:1044-1056, 1059-1137) -- and EOS at the
very first step, its SyntaxError (:1049-1052).  fp32 engine mode: token ids bit-identical.

First run on hardware by the round-5 driver (4 cases passed); a plain member of the GPU suite since round 6."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import valle_amd  # noqa: E402
from oracle import valle_oracle as vo  # noqa: E402
from tests.golden_util import GOLDEN_DIR  # noqa: E402

DEV = "cuda:0"
EOS_DIR = os.path.join(GOLDEN_DIR, "eos")


@pytest.mark.parametrize("name", sorted(f[:-4] for f in os.listdir(EOS_DIR) if f.endswith(".npz")))
def test_fp32_stop_on_eos_matches_the_reference_golden(name):
    z = np.load(os.path.join(EOS_DIR, name + ".npz"))
    cfg = vo.OracleConfig(**{k[4:]: z[k].item() for k in z.files if k.startswith("cfg_")})
    sd = dict(vo.make_state_dict(cfg, int(z["wseed"])))
    w = sd["ar_predict_layer.weight"].clone()
    w[1024] *= float(z["eos_scale"])
    sd["ar_predict_layer.weight"] = w
    x, xl, y = vo.make_inputs(int(z["S"]), int(z["P"]), int(z["iseed"]), Q=cfg.num_quantizers)
    en = torch.tensor([int(z["enroll"])], dtype=torch.int32).to(DEV) if int(z["enroll"]) >= 0 else None
    m = valle_amd.VALLE(cfg.d_model, cfg.nhead, cfg.num_layers, prefix_mode=cfg.prefix_mode, share_embedding=cfg.share_embedding,
                        prepend_bos=cfg.prepend_bos, num_quantizers=cfg.num_quantizers, engine_dtype="fp32")
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).eval()
    if int(z["syntax_error"]):
        with pytest.raises(SyntaxError, match="well trained model shouldn't reach here"):
            m.inference(x.to(DEV), xl.to(DEV), y.to(DEV), en, top_k=1)
        return
    codes = m.inference(x.to(DEV), xl.to(DEV), y.to(DEV), en, top_k=1).cpu()
    want = torch.from_numpy(z["codes"].astype(np.int64))[None]
    assert codes.shape == want.shape, (codes.shape, want.shape)
    assert torch.equal(codes, want), f"{(codes != want).sum().item()} of {want.numel()} token ids differ"
