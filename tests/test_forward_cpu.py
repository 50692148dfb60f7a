"""CPU: the oracle's teacher-forced forward (oracle/valle_oracle.py forward) against the UNMODIFIED
reference's VALLE.forward outputs (tests/golden/forward/*.npz, oracle/make_golden_forward.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import valle_oracle as vo
from oracle.make_golden_forward import CASES, make_batch, make_prompts
from tests.golden_util import GOLDEN_DIR


def load_forward_case(name):
    z = np.load(os.path.join(GOLDEN_DIR, "forward", f"{name}.npz"))
    cfg = vo.OracleConfig(**{k[4:]: z[k].item() for k in z.files if k.startswith("cfg_")})
    padded = "x_lens" in z.files and len(z["x_lens"]) > 0
    x, xl, y, yl = make_batch(int(z["N"]), int(z["S"]), int(z["T"]), int(z["seed"]),
                              [int(v) for v in z["x_lens"]] if padded else None, [int(v) for v in z["y_lens"]] if padded else None)
    kw = dict(train_stage=int(z["train_stage"]))
    if int(z["nar_stage"]) >= 1:
        kw.update(nar_stage=int(z["nar_stage"]), prefix_len=int(z["prefix_len"]))
        if cfg.prefix_mode == 2:
            kw.update(prompt_starts=[int(v) for v in z["starts"]])
        if cfg.prefix_mode == 4:
            kw.update(y_prompts=make_prompts(int(z["N"]), int(z["P4"]), int(z["seed"])))
    return z, cfg, vo.make_state_dict(cfg, 0), x, xl, y, yl, kw


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_forward_matches_reference(name):
    z, cfg, sd, x, xl, y, yl, kw = load_forward_case(name)
    loss, metrics = vo.forward(sd, cfg, x, xl, y, yl, **kw)
    assert abs(float(loss) - float(z["loss"])) <= 2e-5 * abs(float(z["loss"])), (float(loss), float(z["loss"]))
    if float(z["ar_top10"]) >= 0:
        assert abs(metrics["ArTop10Accuracy"] - float(z["ar_top10"])) < 1e-4
    if float(z["nar_top10"]) >= 0:
        assert abs(metrics["NarTop10Accuracy"] - float(z["nar_top10"])) < 1e-4
