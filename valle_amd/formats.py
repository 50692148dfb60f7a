"""Data formats on either side of the decode path: what ``valle/bin/infer.py`` reads
before it can call ``model.inference()``.

* the icefall checkpoint ``.pt``: ONE dict holding ``"model"`` (the state dict), the optimizer / scheduler /
  sampler states, and the training ``params`` flattened into it -- among them the model hyper-parameters
  ``get_model`` consumes and the ``text_tokens`` path (valle/bin/infer.py:126-144, valle/bin/trainer.py:464-475);
* ``unique_text_tokens.k2symbols``: one ``<symbol> <id>`` pair per line (valle/utils/symbol_table.py:76-131);
* the phoneme-id assignment of ``TextTokenCollater``: ``<pad>``=0, ``<bos>``=1, ``<eos>``=2, then the symbols in
  sorted order (valle/data/collation.py:30-57) -- the ids ``ar_text_embedding`` was trained on, so they must be
  reproduced exactly.

Host-side only (strings and ints); nothing here computes on tensors.
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import torch

from .model import get_model

# defaults of add_model_arguments (valle/models/__init__.py:18-95) for keys an old checkpoint may lack
_MODEL_DEFAULTS = dict(model_name="VALL-E", decoder_dim=1024, nhead=16, num_decoder_layers=12, scale_factor=1.0, norm_first=True,
                       add_prenet=False, prefix_mode=0, share_embedding=True, prepend_bos=False, num_quantizers=8)


class Params(dict):
    """dict with attribute access, like icefall.utils.AttributeDict (what infer.py wraps the checkpoint in)."""

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key) from None

    def __setattr__(self, key, value):
        self[key] = value


def read_symbol_table(path: str) -> Dict[str, int]:
    """``<symbol> <id>`` lines -> {symbol: id}; blank lines skipped; duplicate symbols or ids are an error
    (valle/utils/symbol_table.py:76-107)."""
    sym2id: Dict[str, int] = {}
    seen_ids = set()
    with open(path, "r", encoding="utf-8") as f:
        for lineno, line in enumerate(f.read().strip().split("\n"), 1):
            fields = line.split()
            if not fields:
                continue
            if len(fields) != 2:
                raise ValueError(f"{path}:{lineno}: expected '<symbol> <id>', got {len(fields)} fields")
            sym, idx = fields[0], int(fields[1])
            if sym in sym2id:
                raise ValueError(f"{path}:{lineno}: duplicated symbol {sym}")
            if idx in seen_ids:
                raise ValueError(f"{path}:{lineno}: duplicated id {idx}")
            sym2id[sym] = idx
            seen_ids.add(idx)
    return sym2id


def symbol_list(sym2id: Dict[str, int]) -> List[str]:
    """``SymbolTable.symbols`` (symbol_table.py:281-288): the symbols in sorted (string) order -- the file's own
    ids play no role in the collater's id assignment, and ``<eps>`` is an ordinary entry of the inventory."""
    return sorted(sym2id)


class TextTokenCollater:
    """Phoneme strings -> padded id batches (valle/data/collation.py:10-109).

    ids: ``pad_symbol`` 0, then ``bos_symbol`` / ``eos_symbol`` when enabled, then ``sorted(text_tokens)``."""

    def __init__(self, text_tokens: Iterable[str], add_eos: bool = True, add_bos: bool = True, pad_symbol: str = "<pad>",
                 bos_symbol: str = "<bos>", eos_symbol: str = "<eos>"):
        self.pad_symbol, self.bos_symbol, self.eos_symbol = pad_symbol, bos_symbol, eos_symbol
        self.add_eos, self.add_bos = add_eos, add_bos
        vocabulary = [pad_symbol] + ([bos_symbol] if add_bos else []) + ([eos_symbol] if add_eos else []) + sorted(text_tokens)
        self.token2idx = {tok: i for i, tok in enumerate(vocabulary)}
        self.idx2token = list(vocabulary)

    def _frame(self, tokens: Sequence[str]) -> List[str]:
        return ([self.bos_symbol] if self.add_bos else []) + list(tokens) + ([self.eos_symbol] if self.add_eos else [])

    def _batch(self, seqs: List[List[str]]) -> Tuple[torch.Tensor, torch.Tensor]:
        lens = [len(s) for s in seqs]
        width = max(lens)
        pad = self.token2idx[self.pad_symbol]
        ids = torch.full((len(seqs), width), pad, dtype=torch.int64)
        for r, s in enumerate(seqs):
            ids[r, : len(s)] = torch.tensor([self.token2idx[t] for t in s], dtype=torch.int64)
        return ids, torch.tensor(lens, dtype=torch.int32)

    def index(self, tokens_list: Sequence[Sequence[str]]) -> Tuple[torch.Tensor, torch.Tensor]:
        """Lists of phoneme tokens -> (ids int64 (B, L), lengths int32 (B,)); unknown tokens are an error
        (collation.py:59-85)."""
        for tokens in tokens_list:
            missing = [t for t in tokens if t not in self.token2idx]
            assert not missing, f"tokens not in the vocabulary: {missing[:5]}"
        return self._batch([self._frame(t) for t in tokens_list])

    def __call__(self, texts: Sequence[Sequence[str]]) -> Tuple[torch.Tensor, torch.Tensor]:
        """Each text is a sequence whose ELEMENTS are tokens (a str is split into characters, collation.py:87-88)."""
        return self._batch([self._frame([p for p in text]) for text in texts])


def get_text_token_collater(text_tokens_file: str) -> TextTokenCollater:
    """valle/data/collation.py:112-118."""
    return TextTokenCollater(symbol_list(read_symbol_table(text_tokens_file)), add_bos=True, add_eos=True)


def load_checkpoint(path: str, device="cpu", engine_dtype: Optional[str] = None):
    """``load_model`` of valle/bin/infer.py:126-144 on the HIP-backed model: returns ``(model, text_tokens)``.

    The file is the trainer's own pickle (it holds ``pathlib`` objects next to the tensors), so it is read with
    ``weights_only=False`` like the reference does -- load only checkpoints you trust."""
    if not path:
        return None
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    if "model" not in ckpt:
        raise KeyError(f"{path}: not an icefall checkpoint (no 'model' entry)")
    params = Params({**_MODEL_DEFAULTS, **{k: v for k, v in ckpt.items() if k != "model"}})
    if engine_dtype is not None:
        params["engine_dtype"] = engine_dtype
    model = get_model(params)
    missing, unexpected = model.load_state_dict(ckpt["model"], strict=True)
    assert not missing and not unexpected
    model.to(device)
    model.eval()
    return model, params.get("text_tokens")


def save_checkpoint(path: str, model: torch.nn.Module, params: Dict, **extra) -> None:
    """The layout icefall's ``save_checkpoint`` writes for valle/bin/trainer.py:464-475: the module states under
    their names, then every entry of ``params`` at top level."""
    ckpt = {"model": {k: v.detach().cpu() for k, v in model.state_dict().items()}, "model_avg": None, "optimizer": None,
            "scheduler": None, "grad_scaler": None, "sampler": None}
    ckpt.update(extra)
    for k, v in params.items():
        assert k not in ckpt, k
        ckpt[k] = v
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(ckpt, path)
