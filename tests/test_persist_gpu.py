"""The persistent batch-1 AR step (valle_amd/csrc/persist.hip, option "persist") against the launch chain it replaces.

The persistent launch runs the chain's own device functions on the same lane <-> element mapping and in the same reduction
orders, so with the chain set to the same decomposition (16 key splits per head, the same keys per lane, the same bf16 rounding
of the rows that travel packed) every logit of a decode must be BIT-IDENTICAL -- a far stronger check than a tolerance: one stale
or torn hand-off anywhere in the 78 in-launch edges of a step shows up as a differing logit.  Parity of the numbers themselves
with the reference is the business of tests/test_engine_gpu.py and tests/test_parity_sizes_gpu.py, which run with the engine's
defaults -- i.e. through this launch -- at BASELINE configs[1]'s own size.

The shipped default adds the folded LayerNorm (persist_mode bit 5: the dot products run on x * gamma, the row statistics are applied
afterwards -- one workgroup barrier per LayerNorm instead of three).  That is the same arithmetic up to fp32 re-association, so it is
held (a) to the three-barrier form of the same launch within a small fraction of the logits' spread and (b) to itself, bit for bit,
under every request schedule and hand-off timing."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import valle_amd  # noqa: E402

DEV = "cuda:0"
CLASSIC, FOLDED, DEFAULT = 0x114, 0x134, 0x174  # persist_mode: three-barrier LayerNorm (bit-identical to the chain) / folded LayerNorm / + bf16 rows and v_dot2c (the default)


def _inputs(S, P, seed=0):
    g = torch.Generator().manual_seed(4321 + seed)
    x = torch.randint(3, 100, (1, S), generator=g, dtype=torch.int64)
    x[0, 0], x[0, -1] = 1, 2
    y = torch.randint(0, 1024, (1, P, 8), generator=g, dtype=torch.int64)
    return x.to(DEV), y.to(DEV)


@pytest.fixture(scope="module")
def c2_model():
    torch.manual_seed(11)
    return valle_amd.VALLE(1024, 16, 12, prefix_mode=1, engine_dtype="bf16").to(DEV).eval()


@pytest.fixture(scope="module")
def eos_model():
    """The same architecture with the EOS row of the predict layer flipped and scaled: decodes stop on EOS (arg-max or draw, valle.py:1044-1046) after a handful of steps."""
    torch.manual_seed(12)
    m = valle_amd.VALLE(1024, 16, 12, prefix_mode=1, engine_dtype="bf16")
    with torch.no_grad():
        m.ar_predict_layer.weight[1024] *= -1.5  # (this init's EOS logit sits at -1 +- 0.3: now it tops the row every few steps)
    return m.to(DEV).eval()


def _decode(eng, X, Y, S, P, steps, opts, top_k=1, seed=0, temperature=1.0, **gen):
    base = {"persist": 0, "persist_pf": 3, "persist_nk": 2, "persist_mode": CLASSIC, "persist_naps": 0x335854, "persist_sample": 1, "persist_steps": 32, "act_bf16": 2, "qa_nsplit": 8, "qa_nk": 4,
            "steps_per_graph": 0, "ignore_eos": 1}
    base.update(opts)
    for k, v in base.items():
        eng.set_option(k, v)
    eng.set_option("trace_ar_logits", 1)
    eng.prefill(X, [S], Y, [P])
    codes, gl = eng.generate(top_k=top_k, seed=seed, max_new=steps, temperature=temperature, **gen)
    return codes[0, : gl[0]].cpu(), eng.fetch_ar_logits()[:, 0].clone()


def _mode_to_act(mode):
    return (2 if mode & 4 else 0) | (1 if mode & 8 else 0)


@pytest.mark.parametrize("nk,pf,mode", [(2, 3, 0x114), (2, 0, 0x114), (2, 3, 0), (2, 0, 0)])
def test_persistent_step_is_bit_identical_to_the_launch_chain(c2_model, nk, pf, mode):
    S, P, steps = 20, 60, 40
    eng = c2_model.engine_for(1, S, P)
    X, Y = _inputs(S, P)
    ref_codes, ref = _decode(eng, X, Y, S, P, steps, {"qa_nsplit": 16, "qa_nk": nk, "act_bf16": _mode_to_act(mode)})
    assert eng.fetch_u32("persist_active") == 0
    got_codes, got = _decode(eng, X, Y, S, P, steps, {"persist": 1, "persist_nk": nk, "persist_pf": pf, "persist_mode": mode,
                                                      "act_bf16": _mode_to_act(mode)})
    assert eng.fetch_u32("persist_active") == 1, "the persistent step did not run"
    assert eng.fetch_u32("persist_fail") == 0, "a wave of the persistent step gave up waiting for a hand-off"
    assert torch.equal(ref_codes, got_codes)
    assert ref.shape == got.shape and torch.equal(ref, got), f"max |dlogit| {(ref - got).abs().max().item():.3e} (must be 0)"


@pytest.mark.parametrize("nk,pf,mode", [(2, 3, 0x134), (2, 0, 0x134)])
def test_folded_layernorm_matches_the_three_barrier_form(c2_model, nk, pf, mode):
    """LN(x) . W[n] = rstd * (sum_k W[n][k] gamma[k] x[k] - mean * sg[n]) + tb[n]: the same numbers as LayerNorm followed by the
    linear layer (valle/modules/transformer.py:57-74, :296-302) up to fp32 re-association -- which the bf16 roundings of the K/V cache
    and of the packed hidden row amplify to at most a few 1e-4 of the logits' spread (measured 6.4e-4 sigma); bar 5e-3 sigma, every
    step over 96 steps.  The folded form is teacher-forced on the three-barrier form's greedy history (a free-running comparison
    measures where the first sub-noise arg-max tie falls, not the arithmetic: with round 5's folded prefill one of the four
    parameter sets met such a tie and diverged); its own arg-max must agree wherever the margin exceeds the bar."""
    S, P, steps = 47, 225, 96
    eng = c2_model.engine_for(1, S, P)
    X, Y = _inputs(S, P, seed=5)
    opts = {"persist": 1, "persist_nk": nk, "persist_pf": pf, "act_bf16": _mode_to_act(mode)}
    ref_codes, ref = _decode(eng, X, Y, S, P, steps, dict(opts, persist_mode=mode & ~32))
    got_codes, got = _decode(eng, X, Y, S, P, 0, dict(opts, persist_mode=mode), forced=ref_codes[None].to(DEV), forced_lens=[ref_codes.numel()])
    assert eng.fetch_u32("persist_ran") == 1 and eng.fetch_u32("persist_fail") == 0
    assert torch.equal(got_codes, ref_codes)  # (the forced history)
    n = min(ref.shape[0], got.shape[0])
    sigma = ref[:n].std().item()
    err = (ref[:n] - got[:n]).abs().max().item()
    assert not torch.equal(ref[:n], got[:n]), "the folded form did not run (identical bits)"
    assert err <= 5e-3 * sigma, f"max |dlogit| {err:.3e} vs sigma {sigma:.3e}"
    top2 = ref[:n].topk(2, dim=-1).values
    safe = (top2[:, 0] - top2[:, 1]) > 2 * 5e-3 * sigma
    own = eng.fetch_sampled()[0, : ref_codes.numel()]
    k = min(n, own.numel())
    assert torch.equal(own[:k][safe[:k]], ref[:k].argmax(-1)[safe[:k]])


@pytest.mark.parametrize("mode", [0x174])
def test_bf16_activation_rows_with_dot2_match_the_fp32_rows(c2_model, mode):
    """persist_mode bit 6 (D2): the operators' input rows sit in LDS as bf16 and the dot products run on v_dot2c_f32_bf16.  For
    linear2 (and, with bit 3, out-proj) the row already travels as bf16 pairs -- the same products in another summation order; the
    in-projection, linear1 and the predict layer round x * gamma to bf16 first, as the batched step's MFMA GEMMs do.  Teacher-forced
    on the fp32-row form's greedy history over 96 steps: every logit within 3 % of the logits' spread (mean 0.5 %), the arg-max equal
    wherever the margin exceeds that, and bit-reproducible run to run."""
    S, P, steps = 47, 225, 96
    eng = c2_model.engine_for(1, S, P)
    X, Y = _inputs(S, P, seed=6)
    opts = {"persist": 1, "persist_nk": 2, "persist_pf": 3, "act_bf16": _mode_to_act(mode)}
    ref_codes, ref = _decode(eng, X, Y, S, P, steps, dict(opts, persist_mode=mode & ~64))
    forced = dict(forced=ref_codes[None].to(DEV), forced_lens=[ref_codes.numel()])
    _, got = _decode(eng, X, Y, S, P, 0, dict(opts, persist_mode=mode), **forced)
    assert eng.fetch_u32("persist_ran") == 1 and eng.fetch_u32("persist_fail") == 0
    own = eng.fetch_sampled()[0, : ref_codes.numel()].clone()
    _, again = _decode(eng, X, Y, S, P, 0, dict(opts, persist_mode=mode), **forced)
    assert torch.equal(got, again)
    n = min(ref.shape[0], got.shape[0])
    sigma = ref[:n].std().item()
    diff = (ref[:n] - got[:n]).abs()
    assert not torch.equal(ref[:n], got[:n]), "the dot2 form did not run (identical bits)"
    print(f"D2 mode {mode:#x}: max |dlogit| {diff.max().item() / sigma:.3%} of sigma, mean {diff.mean().item() / sigma:.4%}")
    assert diff.max().item() <= 3e-2 * sigma and diff.mean().item() <= 5e-3 * sigma
    top2 = ref[:n].topk(2, dim=-1).values
    k = min(n, own.numel())
    safe = ((top2[:, 0] - top2[:, 1]) > 2 * 3e-2 * sigma)[:k]
    assert torch.equal(own[:k][safe], ref[:k].argmax(-1)[safe])
    # the request schedule without the spread (pf = 0) computes the same bits
    _, pf0 = _decode(eng, X, Y, S, P, 0, dict(opts, persist_mode=mode, persist_pf=0), **forced)
    assert torch.equal(got, pf0)


@pytest.fixture(scope="module")
def c2_model_fp8w():
    torch.manual_seed(13)
    return valle_amd.VALLE(1024, 16, 12, prefix_mode=1, engine_dtype="fp8w").to(DEV).eval()


@pytest.mark.parametrize("pf", [3, 0])
def test_fp8w_persistent_step_is_bit_identical_to_the_fp8w_launch_chain(c2_model_fp8w, pf):
    """FP8W (e4m3fn weight rows + one power-of-two scale per row, BASELINE configs[4]'s weight format): the persistent launch widens
    the codes with the chain's own device function and applies the scale as the chain does (fmaf(dot, scale, bias)), so in the
    three-barrier LayerNorm form every logit of a decode equals the fp8w chain's (16 key splits, 2 keys per lane, bf16 hidden row)."""
    S, P, steps = 20, 60, 40
    eng = c2_model_fp8w.engine_for(1, S, P)
    X, Y = _inputs(S, P)
    ref_codes, ref = _decode(eng, X, Y, S, P, steps, {"qa_nsplit": 16, "qa_nk": 2, "act_bf16": 2})
    assert eng.fetch_u32("persist_ran") == 0
    got_codes, got = _decode(eng, X, Y, S, P, steps, {"persist": 1, "persist_nk": 2, "persist_pf": pf, "persist_mode": CLASSIC, "act_bf16": 2})
    assert eng.fetch_u32("persist_ran") == 1 and eng.fetch_u32("persist_fail") == 0
    assert torch.equal(ref_codes, got_codes)
    assert ref.shape == got.shape and torch.equal(ref, got), f"max |dlogit| {(ref - got).abs().max().item():.3e} (must be 0)"


def test_fp8w_persistent_step_default_form_matches_the_three_barrier_form(c2_model_fp8w):
    """The fp8w engine's default form (folded LayerNorm; the v_dot2c forms need bf16 weights and are masked off) against the
    three-barrier form, teacher-forced over 96 steps at the benchmark's prompt: within 5e-3 of the logits' spread; and it is what an
    fp8w engine runs without any option."""
    S, P, steps = 47, 225, 96
    eng = c2_model_fp8w.engine_for(1, S, P)
    X, Y = _inputs(S, P, seed=5)
    ref_codes, ref = _decode(eng, X, Y, S, P, steps, {"persist": 1, "persist_mode": CLASSIC})
    forced = dict(forced=ref_codes[None].to(DEV), forced_lens=[ref_codes.numel()])
    _, got = _decode(eng, X, Y, S, P, 0, {"persist": 1, "persist_mode": DEFAULT}, **forced)  # bit 6 is masked for fp8 weights: = FOLDED
    assert eng.fetch_u32("persist_ran") == 1 and eng.fetch_u32("persist_fail") == 0
    _, got2 = _decode(eng, X, Y, S, P, 0, {"persist": 1, "persist_mode": FOLDED}, **forced)
    assert torch.equal(got, got2)
    n = min(ref.shape[0], got.shape[0])
    sigma = ref[:n].std().item()
    err = (ref[:n] - got[:n]).abs().max().item()
    assert not torch.equal(ref[:n], got[:n]) and err <= 5e-3 * sigma, (err, sigma)


@pytest.mark.parametrize("pf", [3, 0])
def test_fp32_persistent_step_is_bit_identical_to_the_fp32_launch_chain(pf):
    """The token-exact mode (fp32 weights, fp32 KV cache, fp32 edges) through the persistent launch: three-barrier LayerNorm, nothing
    packed -- every logit of a decode equals the fp32 launch chain's at the same key split (16 splits, 2 keys per lane), here past
    512 keys so that the attention share runs more than one round of its fp32 cache rows."""
    torch.manual_seed(14)
    m = valle_amd.VALLE(1024, 16, 4, prefix_mode=1, engine_dtype="fp32").to(DEV).eval()
    S, P, steps = 47, 225, 300
    eng = m.engine_for(1, S, P)
    X, Y = _inputs(S, P, seed=8)
    ref_codes, ref = _decode(eng, X, Y, S, P, steps, {"qa_nsplit": 16, "qa_nk": 2})
    assert eng.fetch_u32("persist_ran") == 0
    got_codes, got = _decode(eng, X, Y, S, P, steps, {"persist": 1, "persist_nk": 2, "persist_pf": pf, "persist_mode": DEFAULT})  # masked to the fp32 form
    assert eng.fetch_u32("persist_ran") == 1 and eng.fetch_u32("persist_fail") == 0
    assert torch.equal(ref_codes, got_codes)
    assert ref.shape == got.shape and torch.equal(ref, got), f"max |dlogit| {(ref - got).abs().max().item():.3e} (must be 0)"


def test_persistent_step_past_1024_keys_and_at_full_length(c2_model):
    """BASELINE configs[1]'s own lengths plus a longer text: the context passes 1024 keys, where a workgroup's attention share
    takes a second round of key chunks (16 splits x 64 keys per round at 2 keys per lane)."""
    S, P = 60, 225
    eng = c2_model.engine_for(1, S, P)
    X, Y = _inputs(S, P, seed=1)
    steps = 16 * S + 1  # the reference's own cap (valle/models/valle.py:1047): 961 steps, context up to 1246
    ref_codes, ref = _decode(eng, X, Y, S, P, steps, {"qa_nsplit": 16, "qa_nk": 2})
    got_codes, got = _decode(eng, X, Y, S, P, steps, {"persist": 1})
    assert eng.fetch_u32("persist_fail") == 0
    assert ref_codes.numel() == steps
    assert torch.equal(ref_codes, got_codes)
    assert torch.equal(ref, got), f"first differing step {int((ref != got).any(-1).nonzero()[0])}"


def test_persistent_step_graph_replay_equals_eager_and_sampled_decode_is_reproducible(c2_model):
    S, P, steps = 24, 50, 67  # 67 steps: the eager first step, two 32-step launches and two single-step replays
    eng_g = c2_model.engine_for(1, S, P)
    X, Y = _inputs(S, P, seed=2)
    on = {"persist": 1, "persist_mode": FOLDED}
    c1, l1 = _decode(eng_g, X, Y, S, P, steps, on, top_k=-100, seed=7)
    c2, l2 = _decode(eng_g, X, Y, S, P, steps, on, top_k=-100, seed=7)
    assert torch.equal(c1, c2) and torch.equal(l1, l2), "the same seed must reproduce the sampled decode"
    m2 = valle_amd.VALLE(1024, 16, 12, prefix_mode=1, engine_dtype="bf16", use_graph=False)
    m2.load_state_dict(c2_model.state_dict(), strict=True)
    m2 = m2.to(DEV).eval()
    eng_e = m2.engine_for(1, S, P)
    c3, l3 = _decode(eng_e, X, Y, S, P, steps, on, top_k=-100, seed=7)
    assert eng_e.fetch_u32("persist_fail") == 0
    assert torch.equal(c1, c3) and torch.equal(l1, l3), "graph replay and eager launches must agree bit for bit"
    c4, _ = _decode(eng_g, X, Y, S, P, steps, on, top_k=-100, seed=8)
    assert not torch.equal(c1, c4)


@pytest.mark.parametrize("top_k,temperature,ignore_eos", [(1, 1.0, 1), (-100, 1.0, 1), (50, 0.7, 1), (-100, 1.3, 0), (3, 1.0, 0)])
def test_sampling_inside_the_launch_equals_the_sampling_kernel(c2_model, eos_model, top_k, temperature, ignore_eos):
    """persist_sample = 1 (default): after the predict layer every workgroup gathers the logits and runs the sampling kernel's own
    device code (csrc/sampling_dev.h: topk_sampling valle/models/valle.py:1287-1302, stop rule :1044-1048, next input :1013-1015),
    several AR iterations per launch.  Tokens, per-step logits, generated length and the stopping step must equal the one-step
    launches followed by ar_sample_kernel, for every way of cutting the steps into launches -- including utterances that stop on
    EOS in the middle of a launch (random-init weights draw EOS within a few hundred sampled steps)."""
    S, P, steps = 30, 80, 300
    eng = (c2_model if ignore_eos else eos_model).engine_for(1, S, P)
    X, Y = _inputs(S, P, seed=6)
    common = {"persist": 1, "persist_mode": FOLDED, "ignore_eos": ignore_eos}
    ref_codes, ref = _decode(eng, X, Y, S, P, steps, dict(common, persist_sample=0), top_k=top_k, seed=11, temperature=temperature)
    assert eng.fetch_u32("persist_active") == 1
    for per_launch in (32, 1, 3, 8, 300):
        codes, lg = _decode(eng, X, Y, S, P, steps, dict(common, persist_sample=1, persist_steps=per_launch), top_k=top_k, seed=11,
                            temperature=temperature)
        assert eng.fetch_u32("persist_fail") == 0
        assert codes.numel() == ref_codes.numel(), (per_launch, codes.numel(), ref_codes.numel())
        assert torch.equal(codes, ref_codes), per_launch
        n = ref_codes.numel() + 1  # iterations that ran, the stopping one included (the trace buffer is sized by the ENQUEUED steps)
        assert lg.shape[0] >= n and ref.shape[0] >= n and torch.equal(lg[:n], ref[:n]), per_launch
    if not ignore_eos:
        assert 0 < ref_codes.numel() < steps, "model and seed were chosen to stop on EOS before the cap"


def test_sampling_inside_the_launch_follows_forced_tokens(c2_model):
    """Teacher forcing (the parity hook of vle_ar_generate): the fed history is the given one, the engine's own draws are still
    recorded; an id outside the audio vocabulary is reported like the reference's nn.Embedding IndexError."""
    S, P, steps = 20, 40, 45
    eng = c2_model.engine_for(1, S, P)
    X, Y = _inputs(S, P, seed=7)
    g = torch.Generator().manual_seed(5)
    forced = torch.randint(0, 1024, (1, steps), generator=g, dtype=torch.int64)
    common = {"persist": 1, "persist_mode": FOLDED}
    ref_codes, ref = _decode(eng, X, Y, S, P, 0, dict(common, persist_sample=0), forced=forced, forced_lens=[steps])
    got_codes, got = _decode(eng, X, Y, S, P, 0, dict(common, persist_sample=1, persist_steps=7), forced=forced, forced_lens=[steps])
    assert got_codes.numel() == steps and torch.equal(got_codes, forced[0])
    assert torch.equal(ref_codes, got_codes) and torch.equal(ref, got)
    bad = forced.clone()
    bad[0, 9] = 5000
    with pytest.raises(Exception, match="vocabulary"):
        _decode(eng, X, Y, S, P, 0, dict(common, persist_sample=1, persist_steps=7), forced=bad, forced_lens=[steps])
    got2, _ = _decode(eng, X, Y, S, P, 0, dict(common, persist_sample=1), forced=forced, forced_lens=[steps])  # the engine recovers
    assert torch.equal(got2, forced[0])


def test_model_falls_back_to_the_launch_chain_when_the_persistent_launch_gives_up_and_rearms(capfd):
    """The persistent launch needs the whole GPU; a call in which a wave gave up ends with VLE_EBUSY (the DEVICE counter is what
    decides: option persist_inject_fail sets it as a wave that gives up does).  The model API (VALLE.inference_batch, the seam
    bench.py times) repeats the decode from the prefill; the engine keeps its next batch-1 calls on the launch chain and re-arms
    the persistent launch by itself after the back-off -- a shared GPU costs speed for a while, not the request."""
    torch.manual_seed(5)
    m = valle_amd.VALLE(1024, 16, 2, prefix_mode=1, engine_dtype="bf16").to(DEV).eval()
    S, P = 12, 20
    X, Y = _inputs(S, P, seed=9)
    eng = m.engine_for(1, S, P)
    eng.set_option("ignore_eos", 1)
    lens = torch.tensor([S])
    want = m.inference_batch(X, lens, Y, [P], top_k=1, max_new=24)[0]
    assert eng.fetch_u32("persist_ran") == 1 and eng.fetch_u32("persist_fail") == 0 and eng.fetch_u32("persist_fallbacks") == 0
    # the C ABI's own answer: VLE_EBUSY, and the engine refuses to hand out the void call's state
    eng.set_option("persist_inject_fail", 1)
    eng.prefill(X, [S], Y, [P])
    with pytest.raises(valle_amd._lib.VleError) as ei:
        eng.generate(top_k=1, max_new=24)
    assert ei.value.code == valle_amd._lib.VLE_EBUSY
    assert eng.fetch_u32("persist_fallbacks") == 1 and eng.fetch_u32("persist_backoff") == 2 and eng.fetch_u32("persist_active") == 0
    with pytest.raises(valle_amd._lib.VleError):
        eng.nar(None)  # needs a completed generate
    # the next calls run the chain (the counter of the void call does not poison them), then the persistent launch is back
    for left in (1, 0):
        got = m.inference_batch(X, lens, Y, [P], top_k=1, max_new=24)[0]
        assert eng.fetch_u32("persist_ran") == 0 and eng.fetch_u32("persist_backoff") == left
        assert torch.equal(got[:, 0], want[:, 0]), "greedy first-codebook tokens of the two paths"
    got = m.inference_batch(X, lens, Y, [P], top_k=1, max_new=24)[0]
    assert eng.fetch_u32("persist_ran") == 1 and eng.fetch_u32("persist_fail") == 0
    assert torch.equal(got, want)
    # through the model API: the request survives, the message names the fallback, the back-off doubles while it keeps happening
    capfd.readouterr()
    eng.set_option("persist_inject_fail", 1)
    got = m.inference_batch(X, lens, Y, [P], top_k=1, max_new=24)[0]
    assert "launch chain" in capfd.readouterr().err
    assert torch.equal(got[:, 0], want[:, 0])
    assert eng.fetch_u32("persist_fallbacks") == 2 and eng.fetch_u32("persist_ran") == 0 and eng.fetch_u32("persist_backoff") == 1
    got = m.inference_batch(X, lens, Y, [P], top_k=1, max_new=24)[0]  # back-off 1 -> 0: still the chain
    assert eng.fetch_u32("persist_ran") == 0
    eng.set_option("persist_inject_fail", 1)
    got = m.inference_batch(X, lens, Y, [P], top_k=1, max_new=24)[0]  # re-armed, fails again: the back-off is now 4
    assert eng.fetch_u32("persist_fallbacks") == 3 and eng.fetch_u32("persist_backoff") == 3
    eng.set_option("persist_rearm", 1)
    got = m.inference_batch(X, lens, Y, [P], top_k=1, max_new=24)[0]
    assert eng.fetch_u32("persist_ran") == 1 and eng.fetch_u32("persist_fail") == 0 and torch.equal(got, want)


def test_batched_persistent_launch_gives_up_falls_back_and_rearms(small_batch_model, capfd):
    """The same contract for a call of three utterances on the batched persistent launch: a give-up ends the call with VLE_EBUSY, the
    model API repeats the decode from the prefill on the launch chain, the engine backs off (batches of 2 .. 6 count its back-off down)
    and re-arms the batched launch by itself -- the request survives."""
    m = small_batch_model
    X, Y, S, P = _ragged_batch(3)
    lens = torch.tensor(S, dtype=torch.int32)
    eng = m.engine_for(6, max(S), max(P))
    _engine_defaults(eng)
    eng.set_option("ignore_eos", 1)
    eng.set_option("persist_rearm", 1)
    want = m.inference_batch(X, lens, Y, P, None, top_k=1, max_new=24)
    assert eng.fetch_u32("persist_ran") == 1 and eng.fetch_u32("persist_fail") == 0
    f0 = eng.fetch_u32("persist_fallbacks")
    eng.set_option("persist_inject_fail", 1)
    eng.prefill(X, S, Y, P)
    with pytest.raises(valle_amd._lib.VleError) as ei:
        eng.generate(top_k=1, max_new=24)
    assert ei.value.code == valle_amd._lib.VLE_EBUSY
    assert eng.fetch_u32("persist_fallbacks") == f0 + 1 and eng.fetch_u32("persist_backoff") == 2 and eng.fetch_u32("persist_batch_capable") == 0
    for left in (1, 0):  # two calls on the chain (the first tokens of two free-running bf16 paths agree), then the batched launch is back
        got = m.inference_batch(X, lens, Y, P, None, top_k=1, max_new=24)
        assert eng.fetch_u32("persist_ran") == 0 and eng.fetch_u32("persist_backoff") == left
        assert all(torch.equal(g[:3, 0], w[:3, 0]) for g, w in zip(got, want))
    got = m.inference_batch(X, lens, Y, P, None, top_k=1, max_new=24)
    assert eng.fetch_u32("persist_ran") == 1 and eng.fetch_u32("persist_fail") == 0 and all(torch.equal(g, w) for g, w in zip(got, want))
    capfd.readouterr()
    eng.set_option("persist_inject_fail", 1)
    got = m.inference_batch(X, lens, Y, P, None, top_k=1, max_new=24)  # through the model API: repeated on the chain
    assert "launch chain" in capfd.readouterr().err and eng.fetch_u32("persist_ran") == 0
    assert all(g.shape == w.shape and torch.equal(g[:3, 0], w[:3, 0]) for g, w in zip(got, want))
    eng.set_option("persist_rearm", 1)
    got = m.inference_batch(X, lens, Y, P, None, top_k=1, max_new=24)
    assert eng.fetch_u32("persist_ran") == 1 and all(torch.equal(g, w) for g, w in zip(got, want))


def test_decode_with_caller_buffers_at_the_edge_of_their_mappings():
    """Every caller-owned input of the decode path (text, prompt codes, forced tokens, forced NAR history) in a virtual-memory
    mapping of its own that ENDS (then: starts) at the buffer's last (first) byte, an unmapped page behind (in front): a kernel of
    the path that reads past either end of a caller buffer faults here, deterministically, instead of once in forty fresh boxes.
    (The engine's own buffers get the same treatment from VLE_GUARD_ALLOC=1|2 -- tools/fresh_box_probe.py runs both.)"""
    from valle_amd._lib import guarded_like

    torch.manual_seed(11)
    m = valle_amd.VALLE(1024, 16, 2, prefix_mode=1, engine_dtype="bf16").to(DEV).eval()
    S, P = 13, 21
    X, Y = _inputs(S, P, seed=4)
    eng = m.engine_for(1, S, P)
    eng.set_option("ignore_eos", 1)
    eng.prefill(X, [S], Y, [P])
    want0, gl = eng.generate(top_k=1, max_new=40)
    want = eng.nar(None).clone()
    forced = want0[:, : gl[0]].clone()
    for at_start in (False, True):
        Xg, Yg, Fg, Ng = (guarded_like(t, at_start) for t in (X, Y, forced, want))
        eng.prefill(Xg, [S], Yg, [P])
        got0, gl2 = eng.generate(top_k=1, max_new=40)
        assert gl2 == gl and eng.fetch_u32("persist_ran") == 1 and eng.fetch_u32("persist_fail") == 0
        assert torch.equal(eng.nar(None), want)
        eng.prefill(Xg, [S], Yg, [P])
        got1, gl3 = eng.generate(top_k=1, forced=Fg, forced_lens=[gl[0]])
        assert gl3 == gl and torch.equal(got1[:, : gl[0]], forced)
        assert torch.equal(eng.nar(None, forced=Ng), want)
        eng.set_option("persist", 0)  # the launch chain reads the same caller buffers
        eng.prefill(Xg, [S], Yg, [P])
        got2, gl4 = eng.generate(top_k=1, forced=Fg, forced_lens=[gl[0]])
        assert gl4 == gl and torch.equal(got2[:, : gl[0]], forced)
        eng.set_option("persist", 1)


def test_option_sets_without_an_instantiated_form_run_the_launch_chain(c2_model):
    """Round 6 prune: 4 keys per lane, the request schedules 1 / 2 and the packing modes the ladder dropped are no longer compiled.
    Setting such knobs on a default engine must not fail the call (launch_pstep used to reject the step with VLE_EINVAL):
    the engine decodes on the launch chain, and is back on the persistent launch when the knob returns to a shipped value."""
    S, P, steps = 12, 20, 10
    eng = c2_model.engine_for(1, S, P)
    X, Y = _inputs(S, P, seed=2)
    base = {"persist": 1, "persist_mode": DEFAULT, "persist_naps": -1, "act_bf16": 0}
    want, _ = _decode(eng, X, Y, S, P, steps, base)
    assert eng.fetch_u32("persist_ran") == 1
    for knob in ({"persist_nk": 4}, {"persist_pf": 1}, {"persist_pf": 2}, {"persist_mode": 0x17c}, {"persist_mode": 0x164}, {"persist_mode": 0x120}):
        got, _ = _decode(eng, X, Y, S, P, steps, dict(base, **knob))
        assert eng.fetch_u32("persist_active") == 0 and eng.fetch_u32("persist_ran") == 0, knob
        assert got.numel() == steps
    got, _ = _decode(eng, X, Y, S, P, steps, base)
    assert eng.fetch_u32("persist_ran") == 1 and eng.fetch_u32("persist_fail") == 0 and torch.equal(got, want)


def _ragged_batch(B, seed=5):
    # (the packed prefill of all four stays below 128 rows: from there on the engine picks another GEMM tiling, whose summation order --
    #  hence the cached keys -- differs in the last bf16 bit from the one-utterance prefill the bit-identity tests compare with:
    #  21 + 34 + 20 + 20 + 13 + 17 = 125 rows at six utterances)
    S, P = [9, 14, 11, 12, 8, 10][:B], [12, 20, 9, 8, 5, 7][:B]
    g = torch.Generator().manual_seed(seed)
    X = torch.zeros(B, max(S), dtype=torch.int64)
    Y = torch.zeros(B, max(P), 8, dtype=torch.int64)
    for b in range(B):
        X[b, : S[b]] = torch.randint(3, 100, (S[b],), generator=g)
        X[b, 0], X[b, S[b] - 1] = 1, 2
        Y[b, : P[b]] = torch.randint(0, 1024, (P[b], 8), generator=g)
    return X.to(DEV), Y.to(DEV), S, P


def _engine_defaults(eng):
    """the options other tests of this module leave on a shared engine, back to the engine's defaults"""
    for k, v in {"persist": 1, "persist_batch": 1, "persist_pf": 3, "persist_nk": 2, "persist_mode": DEFAULT, "persist_naps": -1, "persist_sample": 1, "persist_steps": 32,
                 "act_bf16": 2, "qa_nsplit": 8, "qa_nk": 4, "steps_per_graph": 0, "persist_trace": 0}.items():
        eng.set_option(k, v)


def _batch_decode(eng, X, Y, S, P, steps, **kw):
    """prefill + AR loop of the whole batch with the logits trace: (first-codebook tokens per utterance, logits [steps][B][V])"""
    eng.set_option("trace_ar_logits", 1)
    eng.prefill(X, S, Y, P)
    codes, gl = eng.generate(max_new=steps, **kw)
    return [codes[b, : gl[b]].cpu() for b in range(len(S))], eng.fetch_ar_logits().clone()


@pytest.fixture(scope="module")
def small_batch_model():
    torch.manual_seed(21)
    return valle_amd.VALLE(1024, 16, 3, prefix_mode=1, engine_dtype="bf16", max_batch=6).to(DEV).eval()


@pytest.mark.parametrize("B", [2, 3, 4, 5, 6])
def test_batched_persistent_launch_is_bit_identical_to_one_utterance_launches(small_batch_model, B):
    """Round 6: 2 .. 6 utterances share ONE persistent launch (csrc/persist_nb.hip: the weights are requested once per step and
    multiplied with every utterance's row; every edge carries B rows).  Per utterance the arithmetic is the one-utterance launch's
    default form on the same lane <-> element mapping, so every logit and token of utterance b must be BIT-IDENTICAL to a
    one-utterance decode of utterance b alone -- ragged lengths, greedy and sampled (utterance b draws from request b's stream)."""
    m = small_batch_model
    X, Y, S, P = _ragged_batch(B)
    eng = m.engine_for(6, max(S), max(P))
    _engine_defaults(eng)
    eng.set_option("ignore_eos", 1)
    assert eng.fetch_u32("persist_batch_capable") == 6
    steps = 40
    for kw in (dict(top_k=1), dict(top_k=20, temperature=0.9, seed=1234)):
        got, lg = _batch_decode(eng, X, Y, S, P, steps, **kw)
        assert eng.fetch_u32("persist_ran") == 1 and eng.fetch_u32("persist_fail") == 0, "the batched persistent launch did not run"
        assert eng.fetch_u32("persist_sample_active") == 1
        for b in range(B):
            kb = dict(kw)
            if "seed" in kb:
                kb["seed"] = (kb["seed"] + b * 0x9E3779B97F4A7C15) & (2**64 - 1)  # request b's stream as request 0 of a one-utterance call
            one, lg1 = _batch_decode(eng, X[b : b + 1, : S[b]], Y[b : b + 1, : P[b]], S[b : b + 1], P[b : b + 1], steps, **kb)
            assert eng.fetch_u32("persist_ran") == 1 and eng.fetch_u32("persist_fail") == 0
            assert torch.equal(got[b], one[0]), (kw, b)
            # row 0 is the PREFILL's logits row (a different predict-layer kernel for one and for several rows: bf16 rounding of its
            # operand, 5 % sigma bar of tests/test_engine_gpu.py); every row after it comes out of the persistent launches
            n = min(lg.shape[0], lg1.shape[0])
            assert (lg[0, b] - lg1[0, 0]).abs().max().item() <= 0.02 * lg1[0, 0].std().item()
            assert n >= steps and torch.equal(lg[1:n, b], lg1[1:n, 0]), f"utterance {b}: max |dlogit| {(lg[1:n, b] - lg1[1:n, 0]).abs().max().item():.3e} (must be 0)"
    # ... and the launch chain (persist_batch = 0 for 3 / 4 utterances) gives the same first tokens from the same sampling streams (two
    # free-running bf16 paths part at their first sub-noise tie, so only the first tokens are compared)
    eng.set_option("persist", 0)
    try:
        chain, _ = _batch_decode(eng, X, Y, S, P, steps, top_k=20, temperature=0.9, seed=1234)
        assert eng.fetch_u32("persist_ran") == 0
    finally:
        eng.set_option("persist", 1)
    for b in range(B):
        assert chain[b].shape == got[b].shape and torch.equal(chain[b][:3], got[b][:3]), b


def test_batched_persistent_launch_past_1024_keys(small_batch_model):
    """Contexts beyond 1024 keys: a workgroup's attention share of an utterance takes further rounds of key chunks (per utterance, after
    the first round all utterances run in lockstep) -- two utterances of different lengths, one crossing 1024 keys 200 steps before the
    other and both ending near 1 350; every logit and token equal to the one-utterance launch."""
    m = small_batch_model
    S, P = [70, 68], [100, 60]  # (each prefill, alone and packed, is >= 128 rows: the same GEMM family rounds the cached keys)
    g = torch.Generator().manual_seed(77)
    X = torch.zeros(2, max(S), dtype=torch.int64)
    Y = torch.zeros(2, max(P), 8, dtype=torch.int64)
    for b in range(2):
        X[b, : S[b]] = torch.randint(3, 100, (S[b],), generator=g)
        X[b, 0], X[b, S[b] - 1] = 1, 2
        Y[b, : P[b]] = torch.randint(0, 1024, (P[b], 8), generator=g)
    X, Y = X.to(DEV), Y.to(DEV)
    eng = m.engine_for(6, max(S), max(P))
    _engine_defaults(eng)
    eng.set_option("ignore_eos", 1)
    got, lg = _batch_decode(eng, X, Y, S, P, 0, top_k=1)  # to the reference's own caps: 16 S + 1 = 1121 / 1025 frames
    assert eng.fetch_u32("persist_ran") == 1 and eng.fetch_u32("persist_fail") == 0
    assert [int(t.numel()) for t in got] == [16 * S[0] + 1, 16 * S[1] + 1]
    for b in range(2):
        one, lg1 = _batch_decode(eng, X[b : b + 1, : S[b]], Y[b : b + 1, : P[b]], S[b : b + 1], P[b : b + 1], 0, top_k=1)
        assert eng.fetch_u32("persist_ran") == 1 and eng.fetch_u32("persist_fail") == 0
        assert torch.equal(got[b], one[0]), b
        n = one[0].numel()
        assert S[b] + P[b] + n > 1024 + 64
        assert torch.equal(lg[1:n, b], lg1[1:n, 0]), f"utterance {b}: first differing step {int((lg[1:n, b] != lg1[1:n, 0]).any(-1).nonzero()[0]) + 1}"


def test_batched_persistent_launch_utterances_stop_on_eos_at_their_own_steps(eos_model):
    """Utterances of a batch hit EOS (arg-max or draw, valle.py:1044-1046) at different iterations: a stopped utterance stays in the
    launch with a frozen cache slot and nothing of it is stored any more; the launch ends when the last one has stopped.  Lengths and
    tokens of every utterance equal its own one-utterance decode."""
    m = eos_model
    X, Y, S, P = _ragged_batch(6, seed=9)
    eng = m.engine_for(6, max(S), max(P))
    _engine_defaults(eng)
    eng.set_option("ignore_eos", 0)
    lens = torch.tensor(S, dtype=torch.int32)
    for kw in (dict(top_k=1), dict(top_k=-100, temperature=1.3, seed=77)):
        for B in (2, 3, 4, 6):
            got = m.inference_batch(X[:B], lens[:B], Y[:B], P[:B], None, max_new=64, **kw)
            assert m.sequential_timings is None and eng.fetch_u32("persist_ran") == 1 and eng.fetch_u32("persist_fail") == 0
            for b in range(B):
                kb = dict(kw)
                if "seed" in kb:
                    kb["seed"] = (kb["seed"] + b * 0x9E3779B97F4A7C15) & (2**64 - 1)
                one = m.inference_batch(X[b : b + 1, : S[b]], lens[b : b + 1], Y[b : b + 1, : P[b]], [P[b]], None, max_new=64, _allow_empty=True, **kb)[0]
                assert got[b].shape == one.shape and torch.equal(got[b][:, 0], one[:, 0]), (kw, B, b, got[b].shape, one.shape)
        lens_seen = {int(t.shape[0]) for t in got}
    assert len(lens_seen) > 1, f"every utterance stopped at the same step ({lens_seen}): the test did not exercise the staggered stop"


def test_two_utterances_are_decoded_one_after_the_other_where_the_batched_launch_is_off(small_batch_model):
    """Round 6 (profiles/r06_small_batch.json): at two utterances the batched launch chain (369 us per AR step, 41 k tokens/s) is slower
    than ONE utterance on the persistent launch (128 us, 56.7 k), so where the batched persistent launch is not available (here:
    persist_batch = 0; fp32 / fp8-weight engines) VALLE.inference_batch decodes a batch of two one after the other.  Ragged lengths;
    every utterance must equal its own batch-1 decode -- greedy, and sampled: utterance b draws from the stream of request b."""
    m = small_batch_model
    X, Y, S, P = _ragged_batch(2)
    lens = torch.tensor(S, dtype=torch.int32)
    eng = m.engine_for(6, max(S), max(P))
    _engine_defaults(eng)
    eng.set_option("ignore_eos", 1)
    eng.set_option("persist_batch", 0)
    try:
        assert eng.fetch_u32("persist_capable") == 1 and eng.fetch_u32("persist_batch_capable") == 0
        for kw in (dict(top_k=1), dict(top_k=20, temperature=0.9, seed=1234)):
            got = m.inference_batch(X, lens, Y, P, None, max_new=40, **kw)
            assert m.sequential_timings is not None and eng.fetch_u32("persist_ran") == 1 and eng.fetch_u32("persist_fail") == 0
            assert m.sequential_timings["ar_steps"] >= 2 * 39
            for b in range(2):
                kb = dict(kw)
                if "seed" in kb:
                    kb["seed"] = (kb["seed"] + b * 0x9E3779B97F4A7C15) & (2**64 - 1)
                one = m.inference_batch(X[b : b + 1, : S[b]], lens[b : b + 1], Y[b : b + 1, : P[b]], [P[b]], None, max_new=40, **kb)[0]
                assert torch.equal(got[b], one), (kw, b)
        # three utterances with the batched launch off: the chain
        X3, Y3 = torch.cat([X, X[:1]]), torch.cat([Y, Y[:1]])
        m.inference_batch(X3, torch.tensor(S + S[:1], dtype=torch.int32), Y3, P + P[:1], None, top_k=1, max_new=8)
        assert m.sequential_timings is None and eng.fetch_u32("persist_ran") == 0
    finally:
        eng.set_option("persist_batch", 1)
    m.inference_batch(X, lens, Y, P, None, top_k=1, max_new=8)
    assert m.sequential_timings is None and eng.fetch_u32("persist_ran") == 1


def test_persistent_step_is_the_default_where_covered_and_only_there():
    torch.manual_seed(3)
    m = valle_amd.VALLE(1024, 16, 2, prefix_mode=1, engine_dtype="bf16").to(DEV).eval()
    eng = m.engine_for(1, 8, 10)
    X, Y = _inputs(8, 10)
    eng.prefill(X, [8], Y, [10])
    assert eng.fetch_u32("persist_active") == 1
    # fp8 weights (round 5): the same launch on e4m3fn rows + row scales; engine mode fp8 decodes its AR loop the same way
    for dtype in ("fp8w", "fp8"):
        m1 = valle_amd.VALLE(1024, 16, 2, prefix_mode=1, engine_dtype=dtype).to(DEV).eval()
        e1 = m1.engine_for(1, 8, 10)
        e1.prefill(X, [8], Y, [10])
        assert e1.fetch_u32("persist_active") == 1, dtype
        codes, gl = e1.generate(top_k=1, max_new=6)
        assert e1.fetch_u32("persist_ran") == 1 and e1.fetch_u32("persist_fail") == 0 and gl[0] >= 1
    # fp32 (round 5): the token-exact mode too, as the three-barrier form with nothing packed
    m4 = valle_amd.VALLE(1024, 16, 2, prefix_mode=1, engine_dtype="fp32").to(DEV).eval()
    e4 = m4.engine_for(1, 8, 10)
    e4.prefill(X, [8], Y, [10])
    assert e4.fetch_u32("persist_active") == 1
    codes, gl = e4.generate(top_k=1, max_new=6)
    assert e4.fetch_u32("persist_ran") == 1 and e4.fetch_u32("persist_fail") == 0 and gl[0] >= 1
    for dtype, d, h in (("bf16", 512, 8), ("fp32", 512, 8)):
        m2 = valle_amd.VALLE(d, h, 2, prefix_mode=1, engine_dtype=dtype).to(DEV).eval()
        e2 = m2.engine_for(1, 8, 10)
        e2.prefill(X, [8], Y, [10])
        assert e2.fetch_u32("persist_active") == 0, (dtype, d)
        codes, gl = e2.generate(top_k=1, max_new=4)
        assert gl[0] >= 1
    # two utterances (bf16, round 6): the batched persistent launch (persist_nb.hip, 2 .. 6 utterances); seven: the launch chain; fp32 engines: the chain
    m3 = valle_amd.VALLE(1024, 16, 2, prefix_mode=1, engine_dtype="bf16", max_batch=7).to(DEV).eval()
    e3 = m3.engine_for(7, 8, 10)
    X2, Y2 = torch.cat([X, X]), torch.cat([Y, Y])
    e3.prefill(X2, [8, 8], Y2, [10, 10])
    assert e3.fetch_u32("persist_active") == 1
    codes, gl = e3.generate(top_k=1, max_new=6)
    assert e3.fetch_u32("persist_ran") == 1 and e3.fetch_u32("persist_fail") == 0 and min(gl) >= 1
    e3.prefill(torch.cat([X] * 7), [8] * 7, torch.cat([Y] * 7), [10] * 7)
    assert e3.fetch_u32("persist_active") == 0
    codes, gl = e3.generate(top_k=1, max_new=4)
    assert e3.fetch_u32("persist_ran") == 0 and min(gl) >= 1
    e4.reserve(2, 8, 10, 16 * 8 + 1)
    e4.prefill(X2, [8, 8], Y2, [10, 10])
    assert e4.fetch_u32("persist_active") == 0 and e4.fetch_u32("persist_batch_capable") == 0
    # slot mode (continuous batching): every live slot advances on the batched launch chain, one slot included
    e3.slots_begin()
    e3.slots_prefill([0], X, [8], Y, [10])
    assert e3.fetch_u32("persist_active") == 0 and e3.fetch_u32("persist_capable") == 0
    e3.slots_step(4, top_k=1)
    assert e3.fetch_u32("persist_ran") == 0
    # ... and one utterance on the same engine: persistent again (the operand table is rebuilt for the one-utterance cache layout)
    e3.prefill(X, [8], Y, [10])
    assert e3.fetch_u32("persist_active") == 1
    codes, gl = e3.generate(top_k=1, max_new=6)
    assert e3.fetch_u32("persist_fail") == 0 and gl[0] >= 1


def test_persistent_step_repeated_decodes_under_changing_timing_stay_identical(c2_model):
    """The hand-offs are correct by protocol (every payload word carries the step's epoch), not by timing: decodes with every
    request schedule, with the first sweeps timed well, badly (no wait) and very late, must give the same bits -- a timing-
    dependent hand-off bug would show as a differing logit or a give-up."""
    S, P, steps = 30, 100, 60
    eng = c2_model.engine_for(1, S, P)
    X, Y = _inputs(S, P, seed=3)
    n = 0
    for mode in (FOLDED, CLASSIC):
        ref_codes, ref = _decode(eng, X, Y, S, P, steps, {"persist": 1, "persist_mode": mode})
        for rep in range(3 if mode == FOLDED else 1):
            for pf in (3, 0):
                for naps in (0x335854, 0, 0xFFFFFF, 0x0F0F0F, 0x123456, 0x325756, 0xF000F0, 0x00FF00, 0x111111, 0x654321):
                    codes, lg = _decode(eng, X, Y, S, P, steps, {"persist": 1, "persist_mode": mode, "persist_pf": pf, "persist_naps": naps})
                    assert eng.fetch_u32("persist_fail") == 0, (pf, hex(naps))
                    assert torch.equal(ref, lg) and torch.equal(ref_codes, codes), (hex(mode), rep, pf, hex(naps))
                    n += 1
    assert n == 80


def test_batched_persistent_launch_under_changing_timing_stays_identical(c2_model):
    """The batched launch's hand-offs are correct by protocol too (every granule of every utterance's row carries the step's epoch):
    12 layers, 4 utterances, the first sweeps timed well, not at all, very late and unevenly, eager single steps and graph replays of
    32-step launches -- the same bits every time, no give-up."""
    B, S, P, steps = 4, 24, 70, 48
    eng = c2_model.engine_for(4, S, P)
    _engine_defaults(eng)
    eng.set_option("ignore_eos", 1)
    xs, ys = zip(*[_inputs(S, P, seed=40 + b) for b in range(B)])
    X, Y = torch.cat(xs), torch.cat(ys)
    ref_codes, ref = _batch_decode(eng, X, Y, [S] * B, [P] * B, steps, top_k=1)
    assert eng.fetch_u32("persist_ran") == 1 and eng.fetch_u32("persist_fail") == 0
    n = 0
    for spg, psteps in ((0, 32), (0, 5), (1, 32)):  # (steps_per_graph = 1: one-step launches)
        eng.set_option("steps_per_graph", spg)
        eng.set_option("persist_steps", psteps)
        for naps in (-1, 0, 0xFFFFFF, 0x0F0F0F, 0x123456, 0xF000F0, 0x00FF00, 0x654321):
            eng.set_option("persist_naps", naps)
            codes, lg = _batch_decode(eng, X, Y, [S] * B, [P] * B, steps, top_k=1)
            assert eng.fetch_u32("persist_ran") == 1 and eng.fetch_u32("persist_fail") == 0, (spg, psteps, hex(naps & 0xFFFFFF))
            assert torch.equal(ref, lg) and all(torch.equal(a, b) for a, b in zip(ref_codes, codes)), (spg, psteps, hex(naps & 0xFFFFFF))
            n += 1
    _engine_defaults(eng)
    assert n == 24
