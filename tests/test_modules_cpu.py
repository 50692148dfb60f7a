"""CPU: host-side contract of the block API (valle_amd/modules.py) -- state-dict keys equal the
reference modules' (recorded in tests/golden/modules by oracle/make_golden_modules.py), the mask
classifier, loud failure without a ROCm device and on configurations outside the decode path."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import valle_amd
from valle_amd import modules as M
from oracle.make_golden_modules import ENCODER_CASES, build_encoder, prefix_lm_mask

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "modules")


def _keys(npz, field="keys"):
    return [str(k) for k in np.load(os.path.join(GOLD, npz))[field]]


@pytest.mark.parametrize("name", sorted(ENCODER_CASES))
def test_encoder_state_dict_keys_match_reference(name):
    d, nhead, layers, adaptive, *_ = ENCODER_CASES[name]
    enc = build_encoder(M, M, d, nhead, layers, adaptive)
    assert list(enc.state_dict().keys()) == _keys(f"{name}.npz")


def test_small_module_state_dict_keys_match_reference():
    assert list(M.MultiheadAttention(64, 4, batch_first=True).state_dict().keys()) == _keys("mha.npz")
    assert list(M.AdaptiveLayerNorm(64, M.LayerNorm(64)).state_dict().keys()) == _keys("norms.npz", "keys_ada")
    assert list(M.TokenEmbedding(64, 100).state_dict().keys()) == _keys("embedding.npz", "keys_tok")
    assert list(M.SinePositionalEmbedding(64, alpha=True).state_dict().keys()) == ["alpha"]


def test_encoder_layers_are_clones_like_the_reference():
    enc = build_encoder(M, M, 64, 2, 3, False)
    w0 = enc.layers[0].self_attn.in_proj_weight
    for l in enc.layers[1:]:
        assert torch.equal(l.self_attn.in_proj_weight, w0) and l.self_attn.in_proj_weight is not w0  # transformer.py:604-605


def test_classify_attn_mask():
    T = 9
    assert M.classify_attn_mask(None, T) == (0, False)
    assert M.classify_attn_mask(prefix_lm_mask(4, T), T) == (4, True)
    assert M.classify_attn_mask(prefix_lm_mask(0, T), T) == (1, True)  # plain causal: row 0 sees key 0
    fm = torch.zeros(T, T).masked_fill(prefix_lm_mask(3, T), float("-inf"))
    assert M.classify_attn_mask(fm, T) == (3, True)
    bad = prefix_lm_mask(4, T).clone()
    bad[6, 1] = True
    with pytest.raises(NotImplementedError):
        M.classify_attn_mask(bad, T)
    with pytest.raises(NotImplementedError):
        M.classify_attn_mask(torch.zeros(T, T) - 1.0, T)
    with pytest.raises(NotImplementedError):
        M.classify_attn_mask(torch.zeros(2, T, T, dtype=torch.bool), T)


def test_key_padding_mask_shapes_outside_the_collater_pattern_raise():
    """Padded query rows get every valid key of their sequence, which equals the reference only for trailing padding per segment
    ([text | pad | audio | pad]) and bool masks: anything else must raise instead of diverging silently (checked before any kernel)."""
    mha = M.MultiheadAttention(64, 4, batch_first=True)
    B, T, S = 2, 8, 3
    xn = torch.zeros(B * T, 64)
    mask = prefix_lm_mask(S, T)
    hole = torch.zeros(B, T, dtype=torch.bool)
    hole[0, 5] = True  # a padded audio frame followed by valid ones
    with pytest.raises(NotImplementedError, match="trailing"):
        mha._attend(xn, B, T, mask, hole)
    thole = torch.zeros(B, T, dtype=torch.bool)
    thole[1, 0] = True  # a padded text position followed by a valid one
    with pytest.raises(NotImplementedError, match="trailing"):
        mha._attend(xn, B, T, mask, thole)
    fm = torch.zeros(B, T)
    fm[0, -1] = float("-inf")
    with pytest.raises(NotImplementedError, match="bool"):
        mha._attend(xn, B, T, mask, fm)
    empty = torch.zeros(B, T, dtype=torch.bool)
    empty[1] = True
    with pytest.raises(ValueError):
        mha._attend(xn, B, T, mask, empty)
    ok = torch.zeros(B, T, dtype=torch.bool)
    ok[0, 2] = ok[0, 6:] = True  # [text pad | audio pad]: accepted -- the call then needs the GPU
    if not torch.cuda.is_available():
        with pytest.raises((RuntimeError, AssertionError)):
            mha._attend(xn, B, T, mask, ok)


def test_configurations_outside_the_decode_path_raise():
    assert not M.TransformerEncoderLayer(64, 4, norm_first=False).norm_first  # post-norm layers are implemented
    with pytest.raises(NotImplementedError):
        M.TransformerEncoderLayer(64, 4, norm_first=True, activation=F.gelu)
    with pytest.raises(NotImplementedError):
        M.MultiheadAttention(64, 4, kdim=32)
    with pytest.raises(NotImplementedError):
        M.LayerNorm(64, eps=1e-6)


def test_no_cpu_path():
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    x = torch.randn(1, 5, 64)
    enc = build_encoder(M, M, 64, 2, 1, False)
    with pytest.raises(RuntimeError):
        enc((x, None))
    with pytest.raises(RuntimeError):
        M.LayerNorm(64)(x)
    with pytest.raises(RuntimeError):
        M.TokenEmbedding(64, 10)(torch.zeros(1, 3, dtype=torch.int64))
    with pytest.raises(RuntimeError):
        M.SinePositionalEmbedding(64)(x)


def test_valle_is_built_from_block_modules():
    m = valle_amd.VALLE(64, 4, 2, prefix_mode=1, engine_dtype="bf16")
    assert isinstance(m.ar_decoder, M.TransformerEncoder) and isinstance(m.nar_decoder.norm, M.AdaptiveLayerNorm)
    assert isinstance(m.ar_decoder.layers[0].self_attn, M.MultiheadAttention)
    assert all(mod.compute_dtype == "bf16" for mod in m.modules() if isinstance(mod, M._HipModule))
    assert m.ar_decoder.layers[0].linear1.weight.shape == (256, 64)
