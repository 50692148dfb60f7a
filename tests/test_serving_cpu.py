"""Host logic of the continuous batcher (valle_amd/serving.py) against a scripted engine: WHICH utterance sits in which slot, when
finished utterances are harvested, and that every request comes back in request order.  The arithmetic (vle_slots_*) is covered on
the GPU by tests/test_serving_gpu.py; nothing here touches the HIP library.
Reference behaviour being scheduled: one utterance per call, valle/bin/infer.py:223-269; the stop rule valle/models/valle.py:1044-1048."""
import random

import pytest
import torch

from valle_amd.serving import ContinuousBatcher, Request


class ScriptedEngine:
    """The slot API of valle_amd.engine.Engine with a script instead of kernels: request i 'generates' lengths[i] frames, and its
    codes are a (G, Q) tensor filled with a value that identifies the request (read back from the text ids it was prefilled with)."""

    def __init__(self, max_batch, lengths, Q):
        self.device = torch.device("cpu")
        self.B, self.lengths, self.Q = max_batch, lengths, Q
        self.busy = {}    # slot -> [request id, frames generated]
        self.log = []     # ("prefill", slots) / ("step", n, live) / ("harvest", slots)
        self.began = 0

    def slots_begin(self):
        self.began += 1
        self.busy.clear()

    def slots_prefill(self, slots, X, x_lens, Y, y_lens, top_k, temperature, seed):
        assert len(slots) == X.shape[0] == Y.shape[0] == len(x_lens) == len(y_lens)
        assert len(set(slots)) == len(slots)
        for j, s in enumerate(slots):
            assert 0 <= s < self.B and s not in self.busy, f"slot {s} handed out twice"
            rid = int(X[j, 0])                      # the test's requests carry their index as first text id
            assert x_lens[j] == rid % 5 + 2          # ... and a length that depends on it: padding must not leak into x_lens
            assert int(X[j, x_lens[j]:].abs().sum()) == 0, "text padded with something else than zeros"
            assert y_lens[j] == rid % 3 + 1 and int(Y[j, y_lens[j]:].abs().sum()) == 0
            self.busy[s] = [rid, 0]
        self.log.append(("prefill", tuple(slots)))

    def slots_step(self, n, top_k, temperature, seed):
        assert self.busy, "a scheduling round with no live slot"
        done, gl = [1] * self.B, [0] * self.B
        for s, st in self.busy.items():
            st[1] = min(st[1] + n, self.lengths[st[0]])
            done[s] = int(st[1] >= self.lengths[st[0]])
            gl[s] = st[1]
        self.log.append(("step", n, len(self.busy)))
        return done, gl

    def slots_harvest(self, slots, gls, enroll):
        out = []
        for s, g in zip(slots, gls):
            rid, got = self.busy.pop(s)
            assert got == g == self.lengths[rid], "harvested before the utterance finished"
            out.append(torch.full((g, self.Q), rid, dtype=torch.int64))
        self.log.append(("harvest", tuple(slots)))
        self.last_enroll = enroll
        return out


class ScriptedModel:
    def __init__(self, eng, Q=8, prefix_mode=1):
        self.num_quantizers, self.prefix_mode, self._eng = Q, prefix_mode, eng

    def engine_for(self, max_batch, max_text, max_prompt):
        assert max_batch == self._eng.B
        return self._eng


def make_requests(n, Q=8, enroll=False):
    reqs = []
    for i in range(n):
        text = torch.zeros(i % 5 + 2, dtype=torch.int64)
        text[0] = i
        prompt = torch.zeros(i % 3 + 1, Q + 1, dtype=torch.int64)  # one codebook more than the model uses: the batcher must cut it
        reqs.append(Request(text, prompt, enroll_len=(i % 4 + 1) if enroll else None))
    return reqs


@pytest.mark.parametrize("max_batch, steps_per_round, harvest_min, n", [(1, 8, None, 5), (4, 8, 1, 13), (8, 3, 2, 40), (8, 16, None, 7), (64, 8, None, 70)])
def test_every_request_comes_back_in_request_order(max_batch, steps_per_round, harvest_min, n):
    rng = random.Random(max_batch * 1000 + n)
    lengths = [rng.randint(1, 60) for _ in range(n)]
    eng = ScriptedEngine(max_batch, lengths, 8)
    cb = ContinuousBatcher(ScriptedModel(eng), max_batch, 16, 16, steps_per_round=steps_per_round, harvest_min=harvest_min)
    out = cb.decode(make_requests(n))
    assert eng.began == 1 and not eng.busy
    assert len(out) == n
    for i, c in enumerate(out):
        assert c.shape == (lengths[i], 8) and bool((c == i).all()), f"request {i} got somebody else's codes"
    assert cb.stats["admitted"] == n
    assert cb.stats["rounds"] == sum(1 for e in eng.log if e[0] == "step")
    assert cb.stats["harvests"] == sum(1 for e in eng.log if e[0] == "harvest")
    # never more utterances in flight than slots
    assert max(e[2] for e in eng.log if e[0] == "step") <= max_batch


def test_finished_utterances_wait_for_harvest_min_while_requests_are_pending():
    # 4 slots, harvest_min 2: utterance 0 finishes after the first round and must wait (its slot stays taken) until a second one has
    # finished; at the end, with nothing pending, the stragglers are harvested together once nothing runs any more
    lengths = [4, 20, 20, 12, 8, 8]
    eng = ScriptedEngine(4, lengths, 8)
    cb = ContinuousBatcher(ScriptedModel(eng), 4, 16, 16, steps_per_round=4, harvest_min=2)
    out = cb.decode(make_requests(6))
    assert [c.shape[0] for c in out] == lengths
    harvests = [e[1] for e in eng.log if e[0] == "harvest"]
    prefills = [e[1] for e in eng.log if e[0] == "prefill"]
    assert prefills[0] == (0, 1, 2, 3)
    assert harvests[0] == (0, 3), "slot 0 (done after round 1) waits for slot 3 (done after round 3): one NAR batch of two"
    assert prefills[1] == (0, 3), "the two waiting requests take the two freed slots"
    assert set(harvests[1]) == {0, 1, 2, 3} and len(harvests) == 2, "nothing pending: the rest is harvested in one batch when all have stopped"


def test_enroll_lengths_follow_their_utterances_and_are_required():
    lengths = [5, 9, 3]
    eng = ScriptedEngine(2, lengths, 8)
    cb = ContinuousBatcher(ScriptedModel(eng, prefix_mode=2), 2, 16, 16, steps_per_round=4, harvest_min=1)
    reqs = make_requests(3, enroll=True)
    out = cb.decode(reqs)
    assert [c.shape[0] for c in out] == lengths
    assert eng.last_enroll is not None and all(isinstance(v, int) for v in eng.last_enroll)
    with pytest.raises(AssertionError):
        cb.decode(make_requests(3, enroll=False))  # prefix_mode 2 / 4 without enroll_len: valle.py:1068-1079 needs it
    cb1 = ContinuousBatcher(ScriptedModel(ScriptedEngine(2, lengths, 8), prefix_mode=1), 2, 16, 16)
    cb1.decode(make_requests(3))
    assert cb1.eng.last_enroll is None


def test_malformed_requests_are_rejected_before_anything_is_scheduled():
    eng = ScriptedEngine(2, [3], 8)
    cb = ContinuousBatcher(ScriptedModel(eng), 2, 16, 16)
    with pytest.raises(AssertionError):
        cb.decode([Request(torch.zeros(2, 2, dtype=torch.int64), torch.zeros(1, 8, dtype=torch.int64))])
    with pytest.raises(AssertionError):
        cb.decode([Request(torch.zeros(2, dtype=torch.int64), torch.zeros(1, 4, dtype=torch.int64))])  # fewer codebooks than the model's
    assert eng.began == 0 and not eng.log


def test_a_busy_gpu_costs_the_session_not_the_requests():
    """Engines of 2 .. 6 slots step on the batched persistent launch (csrc/persist_nb.hip); when that launch cannot keep the whole GPU
    vle_slots_step answers VLE_EBUSY and the slots' utterances are void.  The batcher puts the utterances in flight back at the front of
    the queue, opens a new session and decodes them again: every request still comes back, in request order; other errors and an
    endless series of busy answers are passed on."""
    from valle_amd import _lib

    class BusyEngine(ScriptedEngine):
        def __init__(self, *a, busy_at=(), code=_lib.VLE_EBUSY):
            super().__init__(*a)
            self.busy_at, self.code, self.steps = set(busy_at), code, 0

        def slots_step(self, n, top_k, temperature, seed):
            self.steps += 1
            if self.steps in self.busy_at:
                raise _lib.VleError(self.code, "scripted")
            return super().slots_step(n, top_k, temperature, seed)

    lengths = [5, 17, 9, 30, 3, 12, 8]
    eng = BusyEngine(3, lengths, 8, busy_at={2, 5})
    cb = ContinuousBatcher(ScriptedModel(eng), 3, max_text=8, max_prompt=4, steps_per_round=4, harvest_min=1)
    out = cb.decode(make_requests(len(lengths)), top_k=1)
    assert [int(o.shape[0]) for o in out] == lengths and all(int(o[0, 0]) == i for i, o in enumerate(out))
    assert cb.stats["busy_restarts"] == 2 and eng.began == 3
    # the first prefill after a restart holds the utterances that were in flight, in request order
    prefills = [e[1] for e in eng.log if e[0] == "prefill"]
    assert prefills[0] == (0, 1, 2) and prefills[1] == (0, 1, 2)
    eng2 = BusyEngine(3, lengths, 8, busy_at={1}, code=_lib.VLE_EHIP)
    with pytest.raises(_lib.VleError):
        ContinuousBatcher(ScriptedModel(eng2), 3, max_text=8, max_prompt=4).decode(make_requests(len(lengths)), top_k=1)
    eng3 = BusyEngine(3, lengths, 8, busy_at=set(range(1, 100)))
    cb3 = ContinuousBatcher(ScriptedModel(eng3), 3, max_text=8, max_prompt=4)
    with pytest.raises(_lib.VleError) as ei:
        cb3.decode(make_requests(len(lengths)), top_k=1)
    assert ei.value.code == _lib.VLE_EBUSY and cb3.stats["busy_restarts"] == cb3.max_busy_restarts
