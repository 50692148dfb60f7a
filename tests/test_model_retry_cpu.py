"""Error behaviour of the decode seam (valle_amd/model.py VALLE.inference_batch) against a scripted engine: what the model API does
with the C ABI's status codes.  The device side of the same story (the give-up counter, VLE_EBUSY, the back-off and the re-arm) is
tests/test_persist_gpu.py::test_model_falls_back_to_the_launch_chain...; nothing here touches the HIP library.
Reference: the AR loop's SyntaxError and EOS print, valle/models/valle.py:1044-1055."""
import pytest
import torch

import valle_amd
from valle_amd import _lib


class ScriptedEngine:
    def __init__(self, fail_codes, G=5, Q=8):
        self.device = torch.device("cpu")
        self.fail_codes, self.G, self.Q = list(fail_codes), G, Q
        self.calls = []

    def prefill(self, x, xl, y, yl):
        self.calls.append("prefill")

    def generate(self, **kw):
        self.calls.append("generate")
        if self.fail_codes:
            code = self.fail_codes.pop(0)
            if code is not None:
                raise _lib.VleError(code, "scripted")
        return None, [self.G]

    def fetch_u32(self, what):
        return {"persist_fallbacks": 1, "persist_backoff": 2}[what]

    def nar(self, enroll):
        self.calls.append("nar")
        return torch.arange(self.G * self.Q, dtype=torch.int64).reshape(1, self.G, self.Q)


@pytest.fixture()
def model():
    torch.manual_seed(0)
    return valle_amd.VALLE(64, 4, 2, prefix_mode=1).eval()


def _decode(m, eng, monkeypatch):
    monkeypatch.setattr(m, "engine_for", lambda *a, **k: eng)
    x = torch.ones(1, 6, dtype=torch.int64)
    y = torch.ones(1, 9, 8, dtype=torch.int64)
    return m.inference_batch(x, torch.tensor([6]), y, [9], top_k=1)


def test_busy_gpu_repeats_the_decode_from_the_prefill_once(model, monkeypatch, capsys):
    eng = ScriptedEngine([_lib.VLE_EBUSY])
    out = _decode(model, eng, monkeypatch)
    assert eng.calls == ["prefill", "generate", "prefill", "generate", "nar"]
    assert out[0].shape == (5, 8)
    cap = capsys.readouterr()
    assert "launch chain" in cap.err and "fallback #1" in cap.err and "after 2 calls" in cap.err
    assert "VALL-E EOS [9 -> 14]" in cap.out  # valle.py:1054


def test_a_second_busy_answer_is_the_callers_problem(model, monkeypatch):
    eng = ScriptedEngine([_lib.VLE_EBUSY, _lib.VLE_EBUSY])
    with pytest.raises(_lib.VleError) as ei:
        _decode(model, eng, monkeypatch)
    assert ei.value.code == _lib.VLE_EBUSY and eng.calls.count("generate") == 2 and "nar" not in eng.calls


@pytest.mark.parametrize("code", [_lib.VLE_EHIP, _lib.VLE_ESTATE, _lib.VLE_EINVAL])
def test_other_errors_are_not_retried(model, monkeypatch, code):
    eng = ScriptedEngine([code])
    with pytest.raises(_lib.VleError) as ei:
        _decode(model, eng, monkeypatch)
    assert ei.value.code == code and eng.calls == ["prefill", "generate"]


@pytest.mark.parametrize("codes", [[_lib.VLE_ENOTOKEN], [_lib.VLE_EBUSY, _lib.VLE_ENOTOKEN]])
def test_no_token_is_the_references_syntax_error(model, monkeypatch, codes):
    eng = ScriptedEngine(codes)
    with pytest.raises(SyntaxError, match="well trained model shouldn't reach here"):  # valle.py:1049-1052
        _decode(model, eng, monkeypatch)
    assert "nar" not in eng.calls


class TwoUtteranceEngine(ScriptedEngine):
    """answers the model's questions about the persistent launches; records the batch of every prefill"""

    def __init__(self, batch_capable, capable=1):
        super().__init__([])
        self.words = {"persist_batch_capable": batch_capable, "persist_capable": capable, "persist_fallbacks": 0, "persist_backoff": 0}
        self.batches = []

    def prefill(self, x, xl, y, yl):
        self.batches.append(len(xl))

    def generate(self, **kw):
        return None, [self.G] * self.batches[-1]

    def fetch_u32(self, what):
        return self.words[what]

    def timings(self):
        return dict(prefill_ms=1.0, ar_ms=2.0, nar_ms=3.0, ar_steps=4.0)

    def nar(self, enroll):
        B = self.batches[-1]
        return torch.arange(B * self.G * self.Q, dtype=torch.int64).reshape(B, self.G, self.Q)


@pytest.mark.parametrize("batch_capable,capable,want", [(4, 1, [2]), (2, 1, [2]), (0, 1, [1, 1]), (0, 0, [2])])
def test_two_utterances_take_the_batched_persistent_launch_where_the_engine_offers_it(model, monkeypatch, batch_capable, capable, want):
    """Round 6 policy of VALLE.inference_batch at two utterances (valle_amd/model.py): ONE call where the engine runs 2 .. 6 utterances on
    the batched persistent launch (csrc/persist_nb.hip); one after the other where only the one-utterance launch exists (the batched
    launch chain is slower than that, profiles/r06_small_batch.json); the batched chain where neither does."""
    eng = TwoUtteranceEngine(batch_capable, capable)
    monkeypatch.setattr(model, "engine_for", lambda *a, **k: eng)
    x = torch.ones(2, 6, dtype=torch.int64)
    y = torch.ones(2, 9, 8, dtype=torch.int64)
    out = model.inference_batch(x, torch.tensor([6, 6]), y, [9, 9], top_k=1)
    assert eng.batches == want and len(out) == 2 and all(o.shape == (5, 8) for o in out)
    assert (model.sequential_timings is not None) == (want == [1, 1])
