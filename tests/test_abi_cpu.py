"""CPU: the C-ABI library loads and exports every symbol include/valle_engine.h declares;
host-side logic that needs no GPU (state-dict contract, argument checks)."""
import os
import re

import pytest
import torch

import valle_amd
from oracle import valle_oracle as vo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "valle_engine.h")).read()
    declared = set(re.findall(r"\b(vle_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"vle_config", "vle_engine"}
    assert len(declared) >= 15
    lib = valle_amd._lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in valle_engine.h but not exported"
    assert declared == set(valle_amd._lib.SIGNATURES), "ctypes SIGNATURES out of sync with the header"


def test_config_struct_layout_matches_header():
    import ctypes as C

    assert C.sizeof(valle_amd._lib.VleConfig) == 4 * 24


@pytest.mark.parametrize("kw", [dict(), dict(prepend_bos=True), dict(num_quantizers=1), dict(share_embedding=False), dict(num_quantizers=6)])
def test_state_dict_contract_same_keys_shapes_and_tying(kw):
    cfg = vo.OracleConfig(d_model=64, nhead=4, num_layers=2, prefix_mode=1, **kw)
    spec = vo.state_dict_spec(cfg)  # asserted equal to the reference's state_dict() in oracle/make_golden.py
    m = valle_amd.VALLE(64, 4, 2, prefix_mode=1, **kw)
    got = m.state_dict()
    assert list(got.keys()) == list(spec.keys())
    for k, shape in spec.items():
        assert tuple(got[k].shape) == tuple(shape), k
    m.load_state_dict(vo.make_state_dict(cfg, 0), strict=True)
    if cfg.share_embedding and cfg.num_quantizers > 2:
        assert m.nar_predict_layers[0].weight is m.nar_audio_embeddings[2].weight


def test_constructor_combinations_outside_the_fused_shape():
    """post-norm / prenet / nar_scale_factor != 1 (valle/tests/valle_test.py:106-133 builds them): the reference's state-dict
    keys and shapes, decoded by the block modules -- the fused engine refuses them."""
    for kw in (dict(norm_first=False), dict(add_prenet=True), dict(nar_scale_factor=0.5), dict(norm_first=False, add_prenet=True, nar_scale_factor=2.0)):
        m = valle_amd.VALLE(64, 4, 2, prefix_mode=1, **kw)
        assert not m.fused
        cfg = vo.OracleConfig(d_model=64, nhead=4, num_layers=2, prefix_mode=1, **kw)
        spec = vo.state_dict_spec(cfg)  # == the reference's key order (asserted against it in oracle/make_golden.py)
        got = m.state_dict()
        assert list(got.keys()) == list(spec.keys())
        for k, shape in spec.items():
            assert tuple(got[k].shape) == tuple(shape), k
        m.load_state_dict(vo.make_state_dict(cfg, 0), strict=True)
        with pytest.raises(RuntimeError):
            m.engine_for(1, 4, 4)
    with pytest.raises(NotImplementedError):
        valle_amd.VALLE(64, 4, 2, norm_first=False, engine_dtype="fp8")


def test_no_cpu_path():
    cfg = vo.OracleConfig(d_model=64, nhead=4, num_layers=1, prefix_mode=1)
    m = valle_amd.VALLE(64, 4, 1, prefix_mode=1)
    m.load_state_dict(vo.make_state_dict(cfg, 0))
    x, xl, y = vo.make_inputs(3, 4)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            m.inference(x, xl, y, None, top_k=1)


def test_sine_table_matches_oracle():
    assert torch.equal(valle_amd.sine_pe(300, 64), vo.sine_pe(300, 64))


def test_get_model_surface():
    from oracle.ref_import import AttributeDict

    p = AttributeDict(model_name="VALL-E", decoder_dim=64, nhead=4, num_decoder_layers=2, norm_first=True, add_prenet=False,
                      prefix_mode=1, share_embedding=True, scale_factor=1.0, prepend_bos=False, num_quantizers=8)
    m = valle_amd.get_model(p)
    assert isinstance(m, valle_amd.VALLE) and m.num_quantizers == 8
    p.model_name = "VALL-F"  # valle/models/__init__.py:99-111
    f = valle_amd.get_model(p)
    assert isinstance(f, valle_amd.VALLF) and not f.fused
    cfg = vo.OracleConfig(d_model=64, nhead=4, num_layers=2, prefix_mode=1, model="vallf")
    assert list(f.state_dict().keys()) == list(vo.state_dict_spec(cfg).keys())  # the reference's keys (multihead_attn, norm3)
    f.load_state_dict(vo.make_state_dict(cfg, 0), strict=True)
    p.model_name = "Transformer"
    with pytest.raises(NotImplementedError):
        valle_amd.get_model(p)


def test_op_tune_knows_the_batched_gemm_knobs_and_rejects_the_rest():
    """vle_op_tune is host-only code (process-global kernel-selection knobs): every knob the header names for the
    batched GEMMs is accepted inside its range -- set to its default here, so nothing changes for later tests -- and an unknown
    name or a value outside the range is an error, not a silent no-op."""
    lib = valle_amd._lib.load()
    for name, default in (("gs_fast", 1), ("gs_nf", 1), ("gs_gran", 0), ("gs_msplit", 1), ("gs_formal", 0), ("g1_shared", 1), ("glds_tail", 1), ("glds_t64", 160)):
        assert lib.vle_op_tune(name.encode(), default) == 0, name
    assert lib.vle_op_tune(b"gs_fast", 2) != 0
    assert lib.vle_op_tune(b"gs_nf", -1) != 0
    assert lib.vle_op_tune(b"no_such_knob", 1) != 0


def test_error_codes_of_the_header_and_the_binding_agree():
    """include/valle_engine.h #defines VLE_OK / VLE_E*: the ctypes binding must carry the same numbers (VLE_EBUSY, round 5, is what
    VALLE.inference recognises a persistent launch that lost the GPU by -- a code, not a message text)."""
    import re

    from valle_amd import _lib

    hdr = open(os.path.join(ROOT, "include", "valle_engine.h")).read()
    codes = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+(VLE_(?:OK|E[A-Z]+))\s+\(?(-?\d+)\)?", hdr)}
    assert {"VLE_OK", "VLE_EINVAL", "VLE_ESTATE", "VLE_EHIP", "VLE_EKEY", "VLE_ENOTOKEN", "VLE_EINDEX", "VLE_EBUSY"} <= set(codes), codes
    for name, value in codes.items():
        assert getattr(_lib, name) == value, (name, value)
    assert codes["VLE_EBUSY"] == -7
