"""CPU: the data formats either side of the decode path (valle_amd/formats.py) against fixtures written by the
reference's own classes (oracle/make_golden_formats.py -> tests/golden/formats): symbol-table file, the phoneme-id
assignment of TextTokenCollater, and an icefall-layout checkpoint holding a reference VALLE."""
import json
import os

import pytest
import torch

import valle_amd
from valle_amd import formats

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "formats")


def test_symbol_table_and_collater_ids_match_reference():
    z = json.load(open(os.path.join(GOLD, "collater.json"), encoding="utf-8"))
    table = formats.read_symbol_table(os.path.join(GOLD, "tokens.k2symbols"))
    assert formats.symbol_list(table) == z["table_symbols"]
    assert sorted(table.values()) == z["table_ids"]
    col = valle_amd.get_text_token_collater(os.path.join(GOLD, "tokens.k2symbols"))
    assert col.idx2token == z["symbols"]
    assert col.idx2token[:3] == ["<pad>", "<bos>", "<eos>"]  # ids 0 / 1 / 2 (collation.py:49-57)
    ids, lens = col.index(z["texts"])
    assert ids.dtype == torch.int64 and lens.dtype == torch.int32
    assert ids.tolist() == z["ids"] and lens.tolist() == z["lens"]
    ids, lens = col(z["str_texts"])
    assert ids.tolist() == z["str_ids"] and lens.tolist() == z["str_lens"]
    with pytest.raises(AssertionError):
        col.index([["t", "not-a-phone"]])


def test_symbol_table_rejects_malformed_files(tmp_path):
    p = tmp_path / "bad.k2symbols"
    p.write_text("a 1\nb 1\n")
    with pytest.raises(ValueError):
        formats.read_symbol_table(str(p))
    p.write_text("a 1\na 2\n")
    with pytest.raises(ValueError):
        formats.read_symbol_table(str(p))
    p.write_text("a 1 extra\n")
    with pytest.raises(ValueError):
        formats.read_symbol_table(str(p))
    p.write_text("a 1\n\n  \nb 2\n")
    assert formats.read_symbol_table(str(p)) == {"a": 1, "b": 2}


def test_icefall_checkpoint_of_a_reference_model_loads_strictly():
    path = os.path.join(GOLD, "ckpt_tiny.pt")
    model, text_tokens = valle_amd.load_checkpoint(path, device="cpu", engine_dtype="bf16")
    assert isinstance(model, valle_amd.VALLE) and not model.training
    assert text_tokens == "data/tokenized/unique_text_tokens.k2symbols"
    assert (model.d_model, model.num_heads, model.num_layers, model.prefix_mode, model.engine_dtype) == (16, 2, 1, 1, "bf16")
    ref_sd = torch.load(path, weights_only=False)["model"]
    sd = model.state_dict()
    assert list(sd) == list(ref_sd)
    for k in sd:
        assert torch.equal(sd[k], ref_sd[k]), k
    assert model.nar_predict_layers[0].weight is model.nar_audio_embeddings[2].weight  # tying survives the load


def test_checkpoint_round_trip(tmp_path):
    m = valle_amd.VALLE(32, 2, 1, prefix_mode=2, prepend_bos=True, num_quantizers=4)
    hp = dict(model_name="valle", decoder_dim=32, nhead=2, num_decoder_layers=1, scale_factor=1.0, norm_first=True, add_prenet=False,
              prefix_mode=2, share_embedding=True, prepend_bos=True, num_quantizers=4, text_tokens="tokens.k2symbols")
    p = str(tmp_path / "exp" / "epoch-1.pt")
    valle_amd.save_checkpoint(p, m, hp)
    m2, tt = valle_amd.load_checkpoint(p)
    assert tt == "tokens.k2symbols" and m2.ar_audio_prepend_bos and m2.num_quantizers == 4
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    with pytest.raises(KeyError):
        torch.save({"not_model": 1}, str(tmp_path / "x.pt"))
        valle_amd.load_checkpoint(str(tmp_path / "x.pt"))
