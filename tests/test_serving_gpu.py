"""GPU: continuous batching (valle_amd/serving.py over vle_slots_*): more requests than slots, ragged text / prompt
lengths, utterances finishing at different steps.  fp32 engine mode => every request's codes must equal the CPU oracle's
(= the reference's) token for token, whatever shared the batch with it; and the dense batch API must agree."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import valle_amd  # noqa: E402
from valle_amd import ContinuousBatcher, Request  # noqa: E402
from oracle import valle_oracle as vo  # noqa: E402
from tests.test_engine_gpu import build_model  # noqa: E402

DEV = "cuda:0"


def _requests(n, seed, smin=3, smax=8, pmin=4, pmax=18):
    g = torch.Generator().manual_seed(seed)
    S = torch.randint(smin, smax + 1, (n,), generator=g).tolist()
    P = torch.randint(pmin, pmax + 1, (n,), generator=g).tolist()
    return [vo.make_inputs(S[i], P[i], seed=500 + i) for i in range(n)]


@pytest.mark.parametrize("max_batch,steps_per_round,harvest_min", [(4, 8, None), (3, 5, 1), (16, 8, 2), (4, 8, 3)])
def test_continuous_batching_equals_per_utterance_oracle(max_batch, steps_per_round, harvest_min):
    cfg = vo.OracleConfig(d_model=128, nhead=2, num_layers=2, prefix_mode=1)
    sd = vo.make_state_dict(cfg, 5)
    ins = _requests(11, 1)
    want = [vo.inference(sd, cfg, x, xl, y, None, top_k=1, kv_cache=True)[0] for x, xl, y in ins]
    assert len({w.shape[0] for w in want}) > 2, "the workload must be ragged in G"
    m = build_model(cfg, sd, "fp32", max_batch=max_batch)
    cb = ContinuousBatcher(m, max_batch, max_text=8, max_prompt=18, steps_per_round=steps_per_round, harvest_min=harvest_min)
    got = cb.decode([Request(x[0], y[0]) for x, _, y in ins], top_k=1)
    for i, (g, w) in enumerate(zip(got, want)):
        assert g.shape == w.shape, (i, g.shape, w.shape)
        assert torch.equal(g.cpu(), w), f"request {i} differs"
    assert cb.stats["admitted"] == len(ins) and cb.stats["harvests"] >= 1
    # a second workload on the same batcher (slots are reusable), then the dense API on the same engine
    ins2 = _requests(5, 2)
    got2 = cb.decode([Request(x[0], y[0]) for x, _, y in ins2], top_k=1)
    for (x, xl, y), g in zip(ins2, got2):
        assert torch.equal(g.cpu(), vo.inference(sd, cfg, x, xl, y, None, top_k=1, kv_cache=True)[0])
    x, xl, y = ins[0]
    assert torch.equal(m.inference(x.to(DEV), xl.to(DEV), y.to(DEV), None, top_k=1).cpu()[0], want[0])


def test_continuous_batching_prefix_mode_2_and_bos():
    cfg = vo.OracleConfig(d_model=64, nhead=4, num_layers=2, prefix_mode=2, prepend_bos=True)
    sd = vo.make_state_dict(cfg, 6)
    ins = _requests(6, 3, smin=6, smax=9)
    enroll = [3, 4, 3, 5, 4, 3]
    want = [vo.inference(sd, cfg, x, xl, y, torch.tensor([en], dtype=torch.int32), top_k=1, kv_cache=True)[0] for (x, xl, y), en in zip(ins, enroll)]
    m = build_model(cfg, sd, "fp32", max_batch=2)
    cb = ContinuousBatcher(m, 2, max_text=9, max_prompt=18, steps_per_round=8)
    got = cb.decode([Request(x[0], y[0], en) for (x, _, y), en in zip(ins, enroll)], top_k=1)
    for g, w in zip(got, want):
        assert torch.equal(g.cpu(), w)


def test_slot_api_state_errors():
    cfg = vo.OracleConfig(d_model=64, nhead=4, num_layers=1, prefix_mode=1)
    m = build_model(cfg, vo.make_state_dict(cfg, 0), "fp32", max_batch=2)
    eng = m.engine_for(2, 6, 8)
    x, xl, y = vo.make_inputs(4, 6)
    with pytest.raises(valle_amd._lib.VleError):
        eng.slots_step(1)  # before slots_begin
    eng.slots_begin()
    eng.slots_prefill([1], x.to(DEV), [4], y.to(DEV), [6])
    with pytest.raises(valle_amd._lib.VleError):
        eng.slots_prefill([1], x.to(DEV), [4], y.to(DEV), [6])  # slot busy
    with pytest.raises(valle_amd._lib.VleError):
        eng.slots_harvest([1], [1])  # not finished
    done, gl = eng.slots_step(8)
    assert done[0] == 1 and gl[0] == 0 and gl[1] == 9  # slot 0 stays free; slot 1: first sample + 8 steps


def test_sampled_decode_is_per_request_reproducible_across_batch_sizes():
    """top_k sampling: a request's RNG stream is keyed on (seed, request index, iteration) -- not on the slot or batch position
    it occupies -- so continuous batching with 2, 3 or 5 slots, any harvest policy, and the dense batch API all return the SAME
    sampled codes per request; requests that share a slot one after the other draw different streams."""
    cfg = vo.OracleConfig(d_model=128, nhead=2, num_layers=2, prefix_mode=1)
    sd = vo.make_state_dict(cfg, 5)
    ins = _requests(7, 4)
    reqs = [Request(x[0], y[0]) for x, _, y in ins]
    outs = []
    for max_batch, spr, hm in [(2, 4, 1), (3, 8, 2), (5, 3, None)]:
        m = build_model(cfg, sd, "fp32", max_batch=max_batch)
        cb = ContinuousBatcher(m, max_batch, max_text=8, max_prompt=18, steps_per_round=spr, harvest_min=hm)
        outs.append([c.cpu() for c in cb.decode(reqs, top_k=5, temperature=0.9, seed=123)])
    for other in outs[1:]:
        for i, (a, b) in enumerate(zip(outs[0], other)):
            assert a.shape == b.shape and torch.equal(a, b), f"request {i}: sampled codes depend on the batch geometry"
    # the dense batch API draws the same streams (request index = batch position)
    m = build_model(cfg, sd, "fp32", max_batch=len(ins))
    S = [int(x.shape[1]) for x, _, _ in ins]
    P = [int(y.shape[1]) for _, _, y in ins]
    X = torch.zeros(len(ins), max(S), dtype=torch.int64); Y = torch.zeros(len(ins), max(P), 8, dtype=torch.int64)
    for i, (x, _, y) in enumerate(ins):
        X[i, : S[i]], Y[i, : P[i]] = x[0], y[0]
    dense = m.inference_batch(X.to(DEV), torch.tensor(S, dtype=torch.int32), Y.to(DEV), P, None, top_k=5, temperature=0.9, seed=123)
    for i in range(len(ins)):
        assert torch.equal(dense[i].cpu(), outs[0][i]), f"request {i}: dense batch and continuous batching sample differently"
    # identical requests get different streams (different request indices), a different seed changes everything
    same = [reqs[0]] * 4
    mm = build_model(cfg, sd, "fp32", max_batch=2)
    got = ContinuousBatcher(mm, 2, max_text=8, max_prompt=18).decode(same, top_k=50, temperature=1.5, seed=7)
    firsts = [g[:, 0].cpu() for g in got]
    assert any(f.shape != firsts[0].shape or not torch.equal(f, firsts[0]) for f in firsts[1:])


def test_single_slot_serving_on_the_fused_batch1_step():
    """One slot (max_batch = 1) at a width the fused batch-1 launches cover (d256-h4: head size 64): requests run one after the other
    through the slot API on the SAME engine -- the AR iteration counter, and with it the epochs of the fused launch's q granules,
    restarts at every admission (vle_slots_prefill clears the granules) -- and every request equals its oracle decode."""
    cfg = vo.OracleConfig(d_model=256, nhead=4, num_layers=2, prefix_mode=1)
    sd = vo.make_state_dict(cfg, 9)
    ins = _requests(5, 3)
    want = [vo.inference(sd, cfg, x, xl, y, None, top_k=1, kv_cache=True)[0] for x, xl, y in ins]
    m = build_model(cfg, sd, "fp32", max_batch=1)
    cb = ContinuousBatcher(m, 1, max_text=8, max_prompt=18, steps_per_round=8)
    got = cb.decode([Request(x[0], y[0]) for x, _, y in ins], top_k=1)
    for i, (g, w) in enumerate(zip(got, want)):
        assert torch.equal(g.cpu(), w), f"request {i} differs"
    # and the dense API afterwards (vle_ar_prefill clears them too)
    for (x, xl, y), w in zip(ins[:2], want[:2]):
        assert torch.equal(m.inference(x.to(DEV), xl.to(DEV), y.to(DEV), None, top_k=1).cpu()[0], w)


def test_small_slot_engines_step_on_the_batched_persistent_launch():
    """Round 6: on engines of 2 .. 6 slots (bf16, d1024-h16) vle_slots_step advances the live slots on the batched persistent launch
    (csrc/persist_nb.hip) -- the slots are its utterances, a free or finished slot is a stopped one, every slot keeps its own
    iteration counter (Philox counter) while the hand-off epochs follow a counter of the session.  More requests than slots, ragged
    lengths, utterances stopping on EOS at their own steps, admissions while the other slots are mid-decode: every request's AR tokens
    equal its own one-utterance decode (the one-utterance persistent launch: the same arithmetic per utterance), greedy and sampled."""
    torch.manual_seed(31)
    m = valle_amd.VALLE(1024, 16, 3, prefix_mode=1, engine_dtype="bf16", max_batch=3)
    with torch.no_grad():
        m.ar_predict_layer.weight[1024] *= -1.5  # EOS tops the row every few steps: ragged G
    m = m.to(DEV).eval()
    ins = _requests(8, 7, smin=4, smax=8, pmin=6, pmax=18)
    reqs = [Request(x[0], y[0]) for x, _, y in ins]
    for max_batch, spr in ((3, 8), (2, 5)):
        cb = ContinuousBatcher(m, max_batch, max_text=8, max_prompt=18, steps_per_round=spr, harvest_min=1)
        eng = cb.eng  # (the second batcher uses 2 of the same engine's 3 slots: the third stays a stopped utterance of the launch)
        for kw in (dict(top_k=1), dict(top_k=30, temperature=1.1, seed=99)):
            got = cb.decode(reqs, **kw)
            assert eng.fetch_u32("persist_ran") == 1 and eng.fetch_u32("persist_fail") == 0, "the slots did not step on the batched persistent launch"
            lens = set()
            for i, (x, xl, y) in enumerate(ins):
                kb = dict(kw)
                if "seed" in kb:
                    kb["seed"] = (kb["seed"] + i * 0x9E3779B97F4A7C15) & (2**64 - 1)  # request i's stream as request 0 of a one-utterance call
                one = m.inference_batch(x.to(DEV), xl, y.to(DEV), [int(y.shape[1])], None, _allow_empty=True, **kb)[0]
                assert got[i].shape == one.shape and torch.equal(got[i][:, 0], one[:, 0]), (kw, i, got[i].shape, one.shape)
                lens.add(int(one.shape[0]))
            assert len(lens) > 2, f"the workload must be ragged in G ({lens})"
        # the launch chain on the same slots (persist = 0) starts every request alike (two free-running bf16 paths part at their first tie)
        eng.set_option("persist", 0)
        try:
            chain = cb.decode(reqs, top_k=1)
            assert eng.fetch_u32("persist_ran") == 0
        finally:
            eng.set_option("persist", 1)
        assert all(torch.equal(c[:2, 0], g[:2, 0]) for c, g in zip(chain, cb.decode(reqs, top_k=1)))


def test_small_slot_engine_survives_a_busy_gpu():
    """The batched persistent launch gives up (option persist_inject_fail: the DEVICE counter, as a wave that gives up sets it): the slot
    session ends with VLE_EBUSY, the batcher re-admits the utterances in flight in a new session, which the engine runs on the launch chain
    (its back-off counts slot sessions); the session after that is on the persistent launch again.  Every request comes back."""
    torch.manual_seed(33)
    m = valle_amd.VALLE(1024, 16, 2, prefix_mode=1, engine_dtype="bf16", max_batch=3).to(DEV).eval()
    ins = _requests(5, 11, smin=4, smax=6, pmin=6, pmax=12)
    reqs = [Request(x[0], y[0]) for x, _, y in ins]
    cb = ContinuousBatcher(m, 3, max_text=8, max_prompt=18, steps_per_round=8, harvest_min=1)
    eng = cb.eng
    eng.set_option("ignore_eos", 1)
    want = cb.decode(reqs, top_k=1)
    assert eng.fetch_u32("persist_ran") == 1 and eng.fetch_u32("persist_fail") == 0 and cb.stats["busy_restarts"] == 0
    eng.set_option("persist_inject_fail", 1)
    got = cb.decode(reqs, top_k=1)
    assert cb.stats["busy_restarts"] == 1 and eng.fetch_u32("persist_ran") == 0, "the re-admitted session runs the launch chain"
    assert all(g.shape == w.shape and torch.equal(g[:2, 0], w[:2, 0]) for g, w in zip(got, want))
    again = cb.decode(reqs, top_k=1)  # the next session: back on the persistent launch, the same bits as the undisturbed run
    assert eng.fetch_u32("persist_ran") == 1 and eng.fetch_u32("persist_fail") == 0 and cb.stats["busy_restarts"] == 1
    assert all(torch.equal(g, w) for g, w in zip(again, want))
