"""TEST INFRASTRUCTURE ONLY -- golden fixtures for valle_amd/formats.py, produced by the UNMODIFIED reference
classes under /root/reference (valle/utils/symbol_table.py, valle/data/collation.py, valle/models get_model):

    python oracle/make_golden_formats.py      # writes tests/golden/formats/*

* tokens.k2symbols + collater.json: a symbol file written by the reference's SymbolTable.to_file and the ids the
  reference's TextTokenCollater assigns to a few phoneme sequences;
* ckpt_tiny.pt: a checkpoint in the layout valle/bin/trainer.py:464-475 saves through icefall's save_checkpoint
  ({"model": state_dict, "model_avg", "optimizer", "scheduler", "grad_scaler", "sampler"} + every entry of `params`
  at top level -- icefall is third-party and not vendored, the layout is restated here), holding a reference VALLE.
"""
import importlib.util
import json
import os
import pathlib
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.ref_import import REFERENCE_ROOT, AttributeDict, import_reference  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "formats")


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REFERENCE_ROOT, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    vm = import_reference()
    st = _load("ref_symbol_table", "valle/utils/symbol_table.py")
    col = _load("ref_collation", "valle/data/collation.py")  # imports valle.utils.SymbolTable (icefall stubbed)
    os.makedirs(OUT, exist_ok=True)

    # a phoneme inventory in training order (not sorted, ids with a gap), as prepare / tokenizer scripts produce it
    phones = ["t", "ə", "n", "ɪ", "s", "d", "l", "k", "ɹ", "m", "z", "ˈa", "_", ",", ".", "aɪ", "ʃ", "ŋ", "θ", "ɔː"]
    table = st.SymbolTable()
    for i, p in enumerate(phones):
        table.add(p, index=None if i != 7 else 40)
    path = os.path.join(OUT, "tokens.k2symbols")
    table.to_file(path)
    collater = col.get_text_token_collater(path)
    texts = [["ð"] * 0 + ["t", "ə", "n", "_", "s", "ɪ", "k", "s"], ["aɪ", "_", "m", "ɪ", "s", "t", ".", "ɔː", "l"], ["ʃ", "ŋ"]]
    ids, lens = collater.index(texts)
    ids2, lens2 = collater(["tən", "s.l,k"])  # str input: split into characters
    json.dump({"symbols": collater.idx2token, "texts": texts, "ids": ids.tolist(), "lens": lens.tolist(),
               "str_texts": ["tən", "s.l,k"], "str_ids": ids2.tolist(), "str_lens": lens2.tolist(),
               "table_symbols": table.symbols, "table_ids": table.ids},
              open(os.path.join(OUT, "collater.json"), "w"), ensure_ascii=False, indent=1)

    torch.manual_seed(0)
    hp = dict(model_name="valle", decoder_dim=16, nhead=2, num_decoder_layers=1, scale_factor=1.0, norm_first=True, add_prenet=False,
              prefix_mode=1, share_embedding=True, prepend_bos=False, num_quantizers=8)
    model = vm.get_model(AttributeDict(hp)).eval()
    params = dict(hp, text_tokens="data/tokenized/unique_text_tokens.k2symbols", exp_dir=pathlib.Path("exp/valle"), cur_epoch=3,
                  batch_idx_train=1234, best_train_loss=1.5, base_lr=0.05, world_size=8)
    ckpt = {"model": model.state_dict(), "model_avg": None, "optimizer": None, "scheduler": None, "grad_scaler": None, "sampler": None}
    for k, v in params.items():
        assert k not in ckpt
        ckpt[k] = v
    torch.save(ckpt, os.path.join(OUT, "ckpt_tiny.pt"))

    # the same layout at a shape the HIP engine runs (d_model % 32 == 0): the GPU test goes file -> engine -> codes
    torch.manual_seed(1)
    hp2 = dict(hp, decoder_dim=32, nhead=2, num_decoder_layers=2)
    model2 = vm.get_model(AttributeDict(hp2)).eval()
    ckpt2 = {"model": model2.state_dict(), "model_avg": None, "optimizer": None, "scheduler": None, "grad_scaler": None, "sampler": None}
    for k, v in dict(params, **hp2).items():
        ckpt2[k] = v
    torch.save(ckpt2, os.path.join(OUT, "ckpt_d32.pt"))
    # and the reference's own decode of a seeded input with that model, so the test needs no reference at run time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import valle_oracle as vo
    x, xl, y = vo.make_inputs(7, 11, seed=42)
    with torch.no_grad():
        codes = model2.inference(x, xl, y, enroll_x_lens=None, top_k=1, temperature=1.0)
    torch.save({"x": x, "x_lens": xl, "y": y, "codes": codes}, os.path.join(OUT, "ckpt_d32_decode.pt"))
    print("wrote", OUT, sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
