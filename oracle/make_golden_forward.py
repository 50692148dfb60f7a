"""TEST INFRASTRUCTURE ONLY -- golden fixtures for the teacher-forced forward, produced by the
UNMODIFIED reference's ``VALLE.forward`` (valle/models/valle.py:762-959) in eval mode on CPU fp32:

    python oracle/make_golden_forward.py      # writes tests/golden/forward/*.npz

The reference draws ``nar_stage`` from ``self.rng`` (:891-895) and, for prefix_mode 1, ``prefix_len`` from
``torch.randint`` (:348-350): both draws are captured (wrapping ``_prepare_prompts`` from outside) and stored, so the
oracle / engine are called with the same values.  The Top10Accuracy metrics come from the stand-in of
oracle/ref_import.py (torchmetrics is not installed): the LOSS is pinned by the reference, the metrics are not."""
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import valle_oracle as vo  # noqa: E402
from oracle.make_golden import build_reference, dec_forward_113  # noqa: E402
from oracle.ref_import import import_reference  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "forward")

CASES = {
    "fwd_pm1_n1": dict(cfg=dict(d_model=64, nhead=4, num_layers=2, prefix_mode=1), N=1, S=7, T=24, train_stage=0, seed=3),
    "fwd_pm0_n2": dict(cfg=dict(d_model=64, nhead=4, num_layers=2, prefix_mode=0), N=2, S=6, T=17, train_stage=0, seed=5),
    "fwd_pm1_ar_only": dict(cfg=dict(d_model=128, nhead=2, num_layers=2, prefix_mode=1), N=2, S=9, T=30, train_stage=1, seed=7),
    "fwd_pm1_nar_only": dict(cfg=dict(d_model=128, nhead=2, num_layers=2, prefix_mode=1), N=3, S=5, T=40, train_stage=2, seed=11),
    # VALLF.forward (valle.py:395-564) under torch 1.13's nn.TransformerDecoder loop (oracle/make_golden.py dec_forward_113)
    "fwd_vallf_pm1_n2": dict(cfg=dict(d_model=64, nhead=4, num_layers=2, prefix_mode=1, model="vallf"), N=2, S=6, T=20, train_stage=0, seed=17),
    "fwd_vallf_postnorm_prenet_pm0": dict(cfg=dict(d_model=64, nhead=2, num_layers=1, prefix_mode=0, norm_first=False, add_prenet=True, model="vallf"),
                                          N=1, S=5, T=15, train_stage=0, seed=19),
    "fwd_pm0_bos": dict(cfg=dict(d_model=64, nhead=4, num_layers=1, prefix_mode=0, prepend_bos=True), N=1, S=5, T=13, train_stage=0, seed=13),
    # prefix_mode 2 (a random stretch of the utterance itself is the prompt, blanked in the target codebook) and 4 (PromptedFeatures)
    "fwd_pm2_n2": dict(cfg=dict(d_model=64, nhead=4, num_layers=2, prefix_mode=2), N=2, S=6, T=28, train_stage=0, seed=23),
    "fwd_pm2_nar_only_n3": dict(cfg=dict(d_model=128, nhead=2, num_layers=1, prefix_mode=2), N=3, S=5, T=41, train_stage=2, seed=29),
    "fwd_pm4_n2": dict(cfg=dict(d_model=64, nhead=4, num_layers=2, prefix_mode=4), N=2, S=7, T=22, train_stage=0, seed=31, P4=9),
    "fwd_vallf_pm2_n1": dict(cfg=dict(d_model=64, nhead=4, num_layers=1, prefix_mode=2, model="vallf"), N=1, S=6, T=24, train_stage=0, seed=37),
    # PADDED batches (the collater's shapes: every utterance padded to the longest): the AR loss sums the padded rows too (:875)
    "fwd_pad_pm1_n3": dict(cfg=dict(d_model=64, nhead=4, num_layers=2, prefix_mode=1), N=3, S=7, T=24, train_stage=0, seed=41,
                           x_lens=[7, 5, 6], y_lens=[24, 17, 21]),
    "fwd_pad_pm0_bos_n2": dict(cfg=dict(d_model=64, nhead=4, num_layers=1, prefix_mode=0, prepend_bos=True), N=2, S=5, T=13, train_stage=0, seed=43,
                               x_lens=[5, 4], y_lens=[9, 13]),
    "fwd_pad_pm2_n2": dict(cfg=dict(d_model=64, nhead=4, num_layers=2, prefix_mode=2), N=2, S=6, T=28, train_stage=0, seed=47,
                           x_lens=[6, 6], y_lens=[28, 20]),
    "fwd_pad_pm4_n2": dict(cfg=dict(d_model=64, nhead=4, num_layers=2, prefix_mode=4), N=2, S=7, T=22, train_stage=0, seed=53, P4=8,
                           x_lens=[4, 7], y_lens=[22, 15]),
    "fwd_pad_postnorm_prenet_pm1_n2": dict(cfg=dict(d_model=64, nhead=2, num_layers=1, prefix_mode=1, norm_first=False, add_prenet=True), N=2, S=6, T=20,
                                           train_stage=0, seed=59, x_lens=[6, 5], y_lens=[14, 20]),
    "fwd_pad_nar_only_pm1_n2": dict(cfg=dict(d_model=128, nhead=2, num_layers=1, prefix_mode=1), N=2, S=5, T=30, train_stage=2, seed=61,
                                    x_lens=[3, 5], y_lens=[30, 22]),
}


def make_prompts(N, P, seed):
    """prefix_mode 4: the prompts half of the PromptedFeatures, (N, P, 8) codes, seeded like the batch."""
    g = torch.Generator().manual_seed(seed * 1000 + 7)
    return torch.randint(0, 1024, (N, P, 8), generator=g, dtype=torch.int64)


def make_batch(N, S, T, seed, x_lens=None, y_lens=None):
    """Unpadded: N utterances of S text tokens / T frames.  Padded (x_lens / y_lens given, max == S / T): utterance b is
    make_inputs(x_lens[b], y_lens[b]) padded with text id 0 (the collater's <pad>) and with JUNK codes 777 (the forward must
    blank padded frames itself, valle.py:811)."""
    xs, ys = [], []
    for b in range(N):
        Sb, Tb = (S, T) if x_lens is None else (int(x_lens[b]), int(y_lens[b]))
        x, _, y = vo.make_inputs(Sb, Tb, seed * 100 + b)
        xs.append(torch.nn.functional.pad(x[0], (0, S - Sb), value=0))
        ys.append(torch.nn.functional.pad(y[0], (0, 0, 0, T - Tb), value=777))
    xl = torch.full((N,), S, dtype=torch.int32) if x_lens is None else torch.tensor(x_lens, dtype=torch.int32)
    yl = torch.full((N,), T, dtype=torch.int32) if y_lens is None else torch.tensor(y_lens, dtype=torch.int32)
    return torch.stack(xs), xl, torch.stack(ys), yl


def main():
    vm = import_reference()
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(4)
    for name, spec in CASES.items():
        cfg = vo.OracleConfig(**spec["cfg"])
        sd = vo.make_state_dict(cfg, 0)
        model = build_reference(vm, cfg, sd)
        x, xl, y, yl = make_batch(spec["N"], spec["S"], spec["T"], spec["seed"], spec.get("x_lens"), spec.get("y_lens"))
        model.rng = random.Random(spec["seed"])
        torch.manual_seed(spec["seed"])
        drawn = {}
        orig = model._prepare_prompts

        def spy(y_, y_lens_, codes_, nar_stage, y_prompts_codes):
            state = model.rng.getstate()  # prefix_mode 2 draws one segment start per utterance from self.rng (:368-369): replay them
            emb, plen = orig(y_, y_lens_, codes_, nar_stage, y_prompts_codes)
            drawn.update(nar_stage=int(nar_stage), prefix_len=int(plen))
            if cfg.prefix_mode == 2:
                r = random.Random()
                r.setstate(state)
                drawn["starts"] = [r.randint(0, int(y_lens_[b]) - int(plen)) for b in range(codes_.shape[0])]
            return emb, plen

        model._prepare_prompts = spy
        orig_fwd = torch.nn.TransformerDecoder.forward
        if cfg.model == "vallf":
            torch.nn.TransformerDecoder.forward = dec_forward_113
        y_in, yl_in = y, yl
        if cfg.prefix_mode == 4:  # y / y_lens arrive as PromptedFeatures (valle/data/input_strategies.py:16-35; valle.py:792-798)
            from valle.data.input_strategies import PromptedFeatures

            pr = make_prompts(spec["N"], spec["P4"], spec["seed"])
            y_in = PromptedFeatures(pr, y)
            yl_in = PromptedFeatures(torch.full((spec["N"],), spec["P4"], dtype=torch.int32), yl)
        try:
            with torch.no_grad():
                _, loss, metrics = model(x, xl, y_in, yl_in, reduction="sum", train_stage=spec["train_stage"])
        finally:
            torch.nn.TransformerDecoder.forward = orig_fwd
        out = dict(loss=np.float64(float(loss)), N=np.int32(spec["N"]), S=np.int32(spec["S"]), T=np.int32(spec["T"]),
                   seed=np.int32(spec["seed"]), train_stage=np.int32(spec["train_stage"]),
                   nar_stage=np.int32(drawn.get("nar_stage", -1)), prefix_len=np.int32(drawn.get("prefix_len", -1)),
                   ar_top10=np.float64(metrics.get("ArTop10Accuracy", -1.0)), nar_top10=np.float64(metrics.get("NarTop10Accuracy", -1.0)),
                   starts=np.asarray(drawn.get("starts", []), dtype=np.int32), P4=np.int32(spec.get("P4", 0)),
                   x_lens=np.asarray(spec.get("x_lens", []), dtype=np.int32), y_lens=np.asarray(spec.get("y_lens", []), dtype=np.int32),
                   torch_version=np.bytes_(torch.__version__))
        for k, v in spec["cfg"].items():
            out[f"cfg_{k}"] = np.asarray(v)
        np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)
        print(name, float(loss), drawn, metrics)


if __name__ == "__main__":
    main()
