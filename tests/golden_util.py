"""Helpers shared by the parity tests: load a committed golden fixture (made by
oracle/make_golden.py from the unmodified reference) and rebuild its weights/inputs."""
import os

import numpy as np
import torch

from oracle import valle_oracle as vo

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def list_cases():
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz"))


def load_case(name):
    z = np.load(os.path.join(GOLDEN_DIR, f"{name}.npz"))
    kw = {k[4:]: z[k].item() for k in z.files if k.startswith("cfg_")}
    cfg = vo.OracleConfig(**kw)
    S, P = int(z["S"]), int(z["P"])
    sd = vo.make_state_dict(cfg, int(z["wseed"]))
    x, x_lens, y = vo.make_inputs(S, P, int(z["iseed"]), Q=cfg.num_quantizers)
    enroll = int(z["enroll"])
    enroll_t = torch.tensor([enroll], dtype=torch.int32) if enroll >= 0 else None
    return dict(
        name=name, z=z, cfg=cfg, sd=sd, x=x, x_lens=x_lens, y=y, enroll=enroll_t,
        mode=bytes(z["mode"]).decode(), top_k=int(z["top_k"]),
        codes=torch.from_numpy(z["codes"].astype(np.int64))[None],
    )


def assert_persistent_launch_ran(eng):
    """The headline path is the persistent batch-1 launch (valle_amd/csrc/persist.hip): a parity claim about it must fail when the
    last AR loop silently ran the launch chain instead (CU count, table mismatch, back-off after VLE_EBUSY) or when a wave gave up."""
    ran, fail, fb = eng.fetch_u32("persist_ran"), eng.fetch_u32("persist_fail"), eng.fetch_u32("persist_fallbacks")
    assert ran == 1 and fail == 0 and fb == 0, f"persistent launch: ran={ran} give-ups={fail} fallbacks={fb}"
