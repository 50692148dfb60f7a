"""Continuous batching over the engine's slot API.

The reference's caller decodes one utterance at a time (valle/bin/infer.py:223-269).  ``ContinuousBatcher`` keeps
``max_batch`` utterances in flight on one GPU: a finished utterance (EOS or the reference's length cap,
valle.py:1044-1048) leaves its slot at the next scheduling round -- its 7 NAR stages run then -- and a waiting request is
prefilled into the free slot while the other slots keep decoding.  Every arithmetic step is the engine's
(vle_slots_prefill / vle_slots_step / vle_slots_harvest); this module only decides WHICH utterance sits in which slot.
What an utterance decodes to is independent of its batch mates: greedy decodes are token-identical to ``VALLE.inference``,
and sampled decodes draw from an RNG stream keyed on (seed, request index, iteration) -- not on the slot -- so they do not
depend on ``max_batch`` / scheduling either and equal ``VALLE.inference_batch`` with the same seed.
Engines of 2 .. 6 slots (bf16, d1024-h16) advance their live slots on the batched persistent launch (csrc/persist_nb.hip): one launch per
``slots_step`` call; if that launch cannot keep the whole GPU the call raises ``VleError(VLE_EBUSY)`` and the session's utterances have to
be admitted again (the engine then runs the launch chain for its next calls).
"""
from __future__ import annotations

from collections import deque
from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch

from . import _lib
from .model import VALLE


@dataclass
class Request:
    """One utterance: phoneme ids (S,) incl. BOS/EOS, prompt codes (P, Q); ``enroll_len`` for prefix_mode 2 / 4."""
    text: torch.Tensor
    prompt: torch.Tensor
    enroll_len: Optional[int] = None


class ContinuousBatcher:
    def __init__(self, model: VALLE, max_batch: int, max_text: int, max_prompt: int, steps_per_round: int = 8,
                 harvest_min: Optional[int] = None):
        """``steps_per_round``: AR steps between two scheduling points (the engine replays its captured multi-step graph
        in between).  ``harvest_min``: finished utterances are collected until that many wait (default max_batch / 8)
        before their NAR stages run and their slots are refilled -- the NAR passes and the prefill are MFMA GEMMs whose
        efficiency grows with the number of packed rows (measured on MI355X, C2: 10 ms for one utterance alone, 3.9 ms
        per utterance in a batch of 64), while a finished slot that waits costs no KV traffic.  When nothing is waiting
        for a slot any more, the remaining utterances are harvested together at the end.  Measured (tools/serve_bench.py,
        192 requests, G from 129 to 753 frames, 64 slots): harvest_min 1 / 8 / 16 / 32 -> 288 / 331 / 328 / 319 k tok/s,
        static batches of 64 in arrival order 309 k."""
        assert max_batch >= 1 and steps_per_round >= 1
        self.model, self.max_batch, self.steps_per_round = model, max_batch, steps_per_round
        self.harvest_min = max(1, max_batch // 8) if harvest_min is None else max(1, int(harvest_min))
        self.eng = model.engine_for(max_batch, max_text, max_prompt)
        self.stats = dict(rounds=0, admitted=0, ar_steps=0, harvests=0, busy_restarts=0)
        self.max_busy_restarts = 8  # sessions restarted after VLE_EBUSY before the error is passed on

    @torch.no_grad()
    def decode(self, requests: Sequence[Request], top_k: int = 1, temperature: float = 1.0, seed: int = 0) -> List[torch.Tensor]:
        """Returns one int64 (G_i, Q) tensor per request, in request order, on the model's device."""
        eng, dev, B = self.eng, self.eng.device, self.max_batch
        Q = self.model.num_quantizers
        need_enroll = self.model.prefix_mode in (2, 4)
        for r in requests:
            assert r.text.dim() == 1 and r.prompt.dim() == 2 and r.prompt.shape[1] >= Q, "text (S,), prompt (P, Q)"
            assert not need_enroll or r.enroll_len is not None, "prefix_mode 2/4 needs enroll_len (valle.py:1068-1079)"
        out: List[Optional[torch.Tensor]] = [None] * len(requests)
        pending = deque(range(len(requests)))
        free = list(range(B))
        live = {}  # slot -> request index
        eng.slots_begin()
        while pending or live:
            if pending and free:
                take = [pending.popleft() for _ in range(min(len(free), len(pending)))]
                slots = [free.pop(0) for _ in take]
                S = max(int(requests[i].text.numel()) for i in take)
                P = max(int(requests[i].prompt.shape[0]) for i in take)
                X = torch.zeros(len(take), S, dtype=torch.int64, device=dev)
                Y = torch.zeros(len(take), max(P, 1), Q, dtype=torch.int64, device=dev)
                for j, i in enumerate(take):
                    X[j, : requests[i].text.numel()] = requests[i].text.to(dev)
                    Y[j, : requests[i].prompt.shape[0]] = requests[i].prompt[:, :Q].to(dev)
                eng.slots_prefill(slots, X, [int(requests[i].text.numel()) for i in take], Y,
                                  [int(requests[i].prompt.shape[0]) for i in take], top_k, temperature, seed)
                for s, i in zip(slots, take):
                    live[s] = i
                self.stats["admitted"] += len(take)
            try:
                done, gl = eng.slots_step(self.steps_per_round, top_k, temperature, seed)
            except _lib.VleError as err:
                # engines of 2 .. 6 slots step on the batched persistent launch, which needs the whole GPU: when it could not keep it
                # (VLE_EBUSY) the session's slots are void.  The requests are not: the utterances in flight go back to the FRONT of the
                # queue (request order is kept) and are decoded again in a new session -- the engine runs the launch chain for its next
                # calls and re-arms the persistent launch by itself.  A shared GPU costs time, not a request.
                if err.code != _lib.VLE_EBUSY or self.stats["busy_restarts"] >= self.max_busy_restarts:
                    raise
                self.stats["busy_restarts"] += 1
                for i in sorted(live.values(), reverse=True):
                    pending.appendleft(i)
                live.clear()
                free = list(range(B))
                eng.slots_begin()
                continue
            self.stats["rounds"] += 1
            self.stats["ar_steps"] += self.steps_per_round
            fin = [s for s in live if done[s]]
            running = len(live) - len(fin)
            if fin and (running == 0 or (pending and len(fin) >= self.harvest_min)):
                enroll = [int(requests[live[s]].enroll_len) for s in fin] if need_enroll else None
                codes = eng.slots_harvest(fin, [gl[s] for s in fin], enroll)
                for s, c in zip(fin, codes):
                    out[live.pop(s)] = c.clone()
                    free.append(s)
                self.stats["harvests"] += 1
        return out  # type: ignore[return-value]
