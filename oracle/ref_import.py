"""TEST INFRASTRUCTURE ONLY -- import the *unmodified* reference from /root/reference.

Used only by ``oracle/make_golden.py`` (fixture generation, in the build container) and by
CPU tests that are skipped when ``/root/reference`` is absent (it does not exist on the
GPU box).  Nothing in the product path (``valle_amd/``) may import this module.

The reference's ``valle`` package imports lhotse / icefall / torchmetrics / encodec /
phonemizer at import time; none are installed and none are on the decode hot path.  We
inject three stub modules *before* importing:

* ``icefall.utils``  -- ``make_pad_mask``, ``AttributeDict``, ``str2bool``
  (used at valle/models/valle.py:21, valle/models/__init__.py:4, valle/utils/__init__.py:3)
* ``torchmetrics.classification`` -- no-op ``MulticlassAccuracy`` / ``BinaryAccuracy``
  (constructed at valle/models/valle.py:157-163, 273-279; only called in training)
* ``valle.data`` / ``valle.data.input_strategies`` -- ``PromptedFeatures``
  (valle/data/input_strategies.py:16-35; imported by valle/models/valle.py:24)
"""
import os
import sys
import types

import torch

REFERENCE_ROOT = os.environ.get("VALLE_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "valle", "models", "valle.py"))


class AttributeDict(dict):
    def __getattr__(self, key):
        if key in self:
            return self[key]
        raise AttributeError(key)

    def __setattr__(self, key, value):
        self[key] = value


def _make_pad_mask(lengths, max_len=0):
    max_len = max(max_len, int(lengths.max()))
    ar = torch.arange(max_len, device=lengths.device)[None, :]
    return ar.expand(lengths.size(0), max_len) >= lengths[:, None]


class _NoMetric(torch.nn.Module):
    """Stand-in for torchmetrics.classification.MulticlassAccuracy / BinaryAccuracy (torchmetrics is not installed).
    For the configuration VALLE constructs (valle.py:157-163: top_k=10, average="micro", multidim_average="global",
    ignore_index=1024) it restates the published definition: the fraction of non-ignored targets that are among the
    top_k classes of the prediction along dim 1.  Any other configuration returns 0 (never called on the decode path)."""

    def __init__(self, num_classes=None, top_k=1, average="micro", multidim_average="global", ignore_index=None, **k):
        super().__init__()
        self.top_k, self.ignore_index = top_k, ignore_index
        self.supported = num_classes is not None and average == "micro" and multidim_average == "global"

    def forward(self, preds=None, target=None, *a, **k):
        if not self.supported or preds is None or target is None or preds.dim() != target.dim() + 1:
            return torch.tensor(0.0)
        top = preds.topk(min(self.top_k, preds.shape[1]), dim=1).indices  # (N, k, ...)
        hit = (top == target.unsqueeze(1)).any(dim=1)
        keep = target != self.ignore_index if self.ignore_index is not None else torch.ones_like(hit)
        return (hit & keep).sum().float() / keep.sum().clamp_min(1).float()


class PromptedFeatures:
    def __init__(self, prompts, features):
        self.prompts, self.features = prompts, features

    def to(self, device):
        return PromptedFeatures(self.prompts.to(device), self.features.to(device))

    def sum(self):
        return self.features.sum()

    @property
    def ndim(self):
        return self.features.ndim

    @property
    def data(self):
        return (self.prompts, self.features)


_IMPORTED = None


def import_reference():
    """Returns the reference's ``valle.models`` module (get_model, VALLE, ...)."""
    global _IMPORTED
    if _IMPORTED is not None:
        return _IMPORTED
    if not reference_available():
        raise RuntimeError(f"reference not present at {REFERENCE_ROOT}")

    iu = types.ModuleType("icefall.utils")
    iu.make_pad_mask = _make_pad_mask
    iu.AttributeDict = AttributeDict
    iu.str2bool = lambda v: str(v).lower() in ("1", "true", "yes", "y", "t")
    icefall = types.ModuleType("icefall")
    icefall.utils = iu
    sys.modules.setdefault("icefall", icefall)
    sys.modules.setdefault("icefall.utils", iu)

    tmc = types.ModuleType("torchmetrics.classification")
    tmc.MulticlassAccuracy = _NoMetric
    tmc.BinaryAccuracy = _NoMetric
    tm = types.ModuleType("torchmetrics")
    tm.classification = tmc
    sys.modules.setdefault("torchmetrics", tm)
    sys.modules.setdefault("torchmetrics.classification", tmc)

    vd = types.ModuleType("valle.data")
    vd.__path__ = []
    vdi = types.ModuleType("valle.data.input_strategies")
    vdi.PromptedFeatures = PromptedFeatures
    vd.input_strategies = vdi
    sys.modules["valle.data"] = vd
    sys.modules["valle.data.input_strategies"] = vdi

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import valle.models as vm  # noqa: E402  (unmodified reference code)

    _IMPORTED = vm
    return vm
