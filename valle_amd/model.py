"""Drop-in model surface: ``get_model(params)`` / ``VALLE`` with the reference's constructor,
state-dict layout and ``inference()`` / ``continual()`` signatures
(valle/models/__init__.py:98-136, valle/models/valle.py:727-760, :961-985, :1139-1156),
backed by the HIP engine.  ``bin/infer.py`` of the reference works unchanged when its
``from valle.models import get_model`` resolves to this module.

The module tree is built from the HIP-backed block modules of ``modules.py`` under the reference's
attribute names, so that ``load_state_dict(ckpt["model"], strict=True)`` (valle/bin/infer.py:135-138)
accepts a reference checkpoint and ``model.ar_decoder(...)`` / ``model.nar_decoder(...)`` keep the
reference's block-level call surface.  For the production shape (norm_first, no prenet, nar_scale_factor = 1) ``inference()``
drives the fused, KV-cached, graph-captured engine; the other constructor combinations of the reference (post-norm layers,
prenets, a NAR decoder of a different width: valle/tests/valle_test.py:106-133 runs them) decode through the same HIP block
modules in the reference's own loop structure (one full-sequence pass per AR step, valle.py:1012-1057) -- slower, still HIP
kernels only: there is no PyTorch or CPU fallback.
"""
from __future__ import annotations

import argparse
import random
from typing import List, Optional, Sequence

import torch
import torch.nn as nn

from .engine import Engine, EngineConfig
from . import _lib
from . import engine as _engine

NUM_TEXT_TOKENS = 512  # valle/models/macros.py:2
NUM_AUDIO_TOKENS = 1024  # valle/models/macros.py:5


# the block modules (reference names, reference state-dict keys, HIP forward): modules.py
from .modules import (AdaptiveLayerNorm, AudioPrenet, LayerNorm, SinePositionalEmbedding, TextPrenet, TokenEmbedding,  # noqa: E402
                      TransformerDecoder, TransformerDecoderLayer, TransformerEncoder, TransformerEncoderLayer, set_compute_dtype)


def _decoder(d: int, nhead: int, num_layers: int, adaptive: bool, norm_first: bool = True) -> TransformerEncoder:
    """valle/models/valle.py:141-152 (AR) / :232-246 (NAR): FFN 4d; the final (Adaptive)LayerNorm only for pre-norm layers."""
    layer = TransformerEncoderLayer(d, nhead, dim_feedforward=d * 4, dropout=0.1, batch_first=True, norm_first=norm_first,
                                    adaptive_layer_norm=adaptive)
    norm = None if not norm_first else (AdaptiveLayerNorm(d, norm=LayerNorm(d)) if adaptive else LayerNorm(d))
    return TransformerEncoder(layer, num_layers=num_layers, norm=norm)


def _cross_decoder(d: int, nhead: int, num_layers: int, adaptive: bool, norm_first: bool = True) -> TransformerDecoder:
    """VALL-F's decoders (valle/models/valle.py:141-152, 232-246 with decoder_cls = nn.TransformerDecoder, decoder_layer_cls =
    TransformerDecoderLayer): self-attention over the audio stream, cross-attention over the text."""
    layer = TransformerDecoderLayer(d, nhead, dim_feedforward=d * 4, dropout=0.1, batch_first=True, norm_first=norm_first,
                                    adaptive_layer_norm=adaptive)
    norm = None if not norm_first else (AdaptiveLayerNorm(d, norm=LayerNorm(d)) if adaptive else LayerNorm(d))
    return TransformerDecoder(layer, num_layers=num_layers, norm=norm)


def _request_seed_base(seed: int, b: int) -> int:
    """ops.topk_sample draws row r from the stream of request r of ``seed``; the block path samples one row (r = 0) per step, so
    utterance b of a batch passes the seed whose request-0 stream IS request b's stream of ``seed``:
    request_seed(s, r) = mix(s + GOLDEN * (r + 1)) (csrc/common.h), hence request_seed(seed + GOLDEN * b, 0) == request_seed(seed, b).
    Sampled decodes of the block path therefore draw exactly what the fused engine and the serving path draw for every b."""
    return (seed + 0x9E3779B97F4A7C15 * b) & (2**64 - 1)


class PromptedFeatures:
    """``valle.data.input_strategies.PromptedFeatures`` (valle/data/input_strategies.py:16-35): the (prompts, features) pair the
    prefix_mode 4 collation hands to ``forward`` as ``y`` -- and, holding the two length vectors, as ``y_lens`` (valle.py:792-798)."""

    def __init__(self, prompts, features):
        self.prompts = prompts
        self.features = features

    def to(self, device):
        return PromptedFeatures(self.prompts.to(device), self.features.to(device))

    def sum(self):
        return self.features.sum()

    @property
    def ndim(self):
        return self.features.ndim

    @property
    def data(self):
        return (self.prompts, self.features)


class VALLE(nn.Module):
    """HIP-backed VALL-E (valle/models/valle.py:722-1238)."""

    _decoder_factory = staticmethod(_decoder)
    _eos_name = "VALL-E"

    def __init__(
        self,
        d_model: int,
        nhead: int,
        num_layers: int,
        norm_first: bool = True,
        add_prenet: bool = False,
        prefix_mode: int = 0,
        share_embedding: bool = True,
        nar_scale_factor: float = 1.0,
        prepend_bos: bool = False,
        num_quantizers: int = 8,
        *,
        engine_dtype: str = "fp32",
        max_batch: int = 1,
        max_text: int = 0,
        max_prompt: int = 0,
        max_gen: int = 0,
        use_graph: bool = True,
        **kwargs,
    ):
        super().__init__()
        assert num_quantizers >= 1
        d = d_model
        nd = int(d_model * nar_scale_factor)                                       # valle.py:83
        nar_heads, nar_layers = int(nhead * nar_scale_factor), int(num_layers * nar_scale_factor)  # :235, :241
        # the fused engine implements the production shape; every other combination decodes through the block modules
        self.fused = type(self)._decoder_factory is _decoder and bool(norm_first) and not add_prenet and nar_scale_factor == 1.0
        if not self.fused and engine_dtype == "fp8":
            raise NotImplementedError("engine_dtype='fp8' (fp8 activations) exists only in the fused engine's packed passes")
        self.norm_first, self.add_prenet, self.nar_scale_factor = bool(norm_first), bool(add_prenet), float(nar_scale_factor)
        self.d_model, self.num_heads, self.num_layers = d_model, nhead, num_layers
        self.prefix_mode, self.num_quantizers = prefix_mode, num_quantizers
        self.ar_audio_prepend_bos = prepend_bos
        self.share_embedding = share_embedding
        self.engine_dtype, self.max_batch = engine_dtype, max_batch
        self.max_text, self.max_prompt, self.max_gen, self.use_graph = max_text, max_prompt, max_gen, use_graph

        self.ar_text_embedding = TokenEmbedding(d, NUM_TEXT_TOKENS)
        self.nar_text_embedding = TokenEmbedding(nd, NUM_TEXT_TOKENS)
        self.ar_audio_embedding = TokenEmbedding(d, NUM_AUDIO_TOKENS + 1 + int(prepend_bos))
        self.ar_text_prenet = TextPrenet(d) if add_prenet else nn.Identity()      # valle.py:99-126
        self.ar_audio_prenet = AudioPrenet(d) if add_prenet else nn.Identity()
        self.ar_text_position = SinePositionalEmbedding(d, dropout=0.1, scale=False, alpha=True)
        self.ar_audio_position = SinePositionalEmbedding(d, dropout=0.1, scale=False, alpha=True)
        self.ar_decoder = self._decoder_factory(d, nhead, num_layers, adaptive=False, norm_first=norm_first)
        self.ar_predict_layer = nn.Linear(d, NUM_AUDIO_TOKENS + 1, bias=False)
        if num_quantizers > 1:
            self.nar_audio_embeddings = nn.ModuleList(
                [TokenEmbedding(nd, NUM_AUDIO_TOKENS + 1)] + [TokenEmbedding(nd, NUM_AUDIO_TOKENS) for _ in range(num_quantizers - 1)]
            )
            self.nar_text_prenet = TextPrenet(nd) if add_prenet else nn.Identity()   # valle.py:182-219
            self.nar_audio_prenet = AudioPrenet(nd) if add_prenet else nn.Identity()
            self.nar_text_position = SinePositionalEmbedding(nd, dropout=0.0, scale=False, alpha=False)
            self.nar_audio_position = SinePositionalEmbedding(nd, dropout=0.1, scale=False, alpha=False)
            self.nar_decoder = self._decoder_factory(nd, nar_heads, nar_layers, adaptive=True, norm_first=norm_first)
            self.nar_predict_layers = nn.ModuleList([nn.Linear(nd, NUM_AUDIO_TOKENS, bias=False) for _ in range(num_quantizers - 1)])
            self.nar_stage_embeddings = nn.ModuleList([TokenEmbedding(nd, 1) for _ in range(num_quantizers - 1)])
            if share_embedding:
                for j in range(0, num_quantizers - 2):  # valle.py:268-271
                    self.nar_predict_layers[j].weight = self.nar_audio_embeddings[j + 2].weight
        self.rng = random.Random(0)  # valle.py:165 (forward() draws nar_stage from it)
        self.requires_grad_(False)
        # the block modules run the same element type as the engine ("fp8": the engine's fp8 activations exist only in
        # its packed passes; the block API then computes like "fp8w": bf16 kernels on W')
        set_compute_dtype(self, "fp8w" if engine_dtype == "fp8" else engine_dtype)
        self._engine: Optional[Engine] = None
        self._engine_key = None

    # ---- engine life cycle --------------------------------------------------------------------
    def _invalidate(self):
        if self._engine is not None:
            self._engine.close()
        self._engine, self._engine_key = None, None

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._invalidate()
        return r

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._invalidate()
        return r

    def engine_for(self, batch: int, text_len: int, prompt_len: int, gen_len: int = 0) -> Engine:
        """Engine sized for the request (rebuilt only when a capacity grows)."""
        if not self.fused:
            raise RuntimeError("this constructor combination (post-norm / prenet / nar_scale_factor != 1) decodes through the block "
                               "modules; the fused engine implements the production shape only")
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("the HIP engine needs the model on a ROCm device: call .to('cuda') first (no CPU path)")
        need_gen = max(gen_len, 16 * text_len + 1)
        if self._engine is not None:
            c = self._engine.cfg
            if self._engine_key == (dev.index or 0, self.engine_dtype):
                if not (c.max_batch >= batch and c.max_text >= text_len and c.max_prompt >= prompt_len and c.max_gen_eff() >= need_gen):
                    # a capacity grew: re-create the buffers only, the weights stay on the device (vle_reserve)
                    try:
                        self._engine.reserve(batch, text_len, prompt_len, need_gen)
                    except Exception:
                        # a failed vle_reserve leaves the engine unusable (VLE_ESTATE on every call): drop it, so the next
                        # request builds a fresh one instead of reusing the broken handle
                        self._invalidate()
                        raise
                    c = self._engine.cfg
                    self.max_text, self.max_prompt, self.max_batch, self.max_gen = c.max_text, c.max_prompt, c.max_batch, c.max_gen
                return self._engine
            self._invalidate()
        max_text = max(self.max_text, text_len)
        max_gen = max(self.max_gen, gen_len)
        if max_gen and max_gen < 16 * max_text + 1:
            max_gen = max(max_gen, need_gen)
        cfg = EngineConfig(
            d_model=self.d_model, nhead=self.num_heads, num_layers=self.num_layers, num_quantizers=self.num_quantizers,
            prefix_mode=self.prefix_mode, prepend_bos=self.ar_audio_prepend_bos, dtype=self.engine_dtype,
            max_batch=max(self.max_batch, batch), max_text=max_text, max_prompt=max(self.max_prompt, prompt_len),
            max_gen=max_gen, device=dev.index or 0, use_graph=self.use_graph,
        )
        eng = Engine(cfg)
        eng.load_state_dict(self.state_dict())
        self._engine, self._engine_key = eng, (dev.index or 0, self.engine_dtype)
        self.max_text, self.max_prompt, self.max_batch, self.max_gen = cfg.max_text, cfg.max_prompt, cfg.max_batch, max_gen
        return eng

    # ---- VALLE.inference (valle/models/valle.py:961-1137) ---------------------------------------
    @torch.no_grad()
    def inference(
        self,
        x: torch.Tensor,
        x_lens: torch.Tensor,
        y: torch.Tensor,
        enroll_x_lens: Optional[torch.Tensor] = None,
        top_k: int = -100,
        temperature: float = 1.0,
        seed: Optional[int] = None,
    ) -> torch.Tensor:
        assert x.ndim == 2, x.shape
        assert x_lens.ndim == 1, x_lens.shape
        assert y.ndim == 3, y.shape
        assert y.shape[0] == 1, y.shape
        assert torch.all(x_lens > 0)
        out = self.inference_batch(x, x_lens, y, [y.shape[1]], enroll_x_lens, top_k, temperature, seed)
        return out[0][None]

    @torch.no_grad()
    def inference_batch(
        self,
        x: torch.Tensor,
        x_lens: torch.Tensor,
        y: torch.Tensor,
        y_lens: Sequence[int],
        enroll_x_lens: Optional[torch.Tensor] = None,
        top_k: int = -100,
        temperature: float = 1.0,
        seed: Optional[int] = None,
        max_new: int = 0,
        _allow_empty: Optional[bool] = None,
    ) -> List[torch.Tensor]:
        """B independent utterances (the reference is batch-1, valle.py:989): returns a list of
        (G_b, Q) int64 tensors on the model's device."""
        B = x.shape[0]
        xl = [int(v) for v in x_lens.tolist()]
        yl = [int(v) for v in y_lens]
        if not self.fused:
            return self._inference_blocks(x, xl, y, yl, enroll_x_lens, top_k, temperature, seed, max_new)
        if seed is None:
            # the reference samples from torch's global generator (valle.py:1301): draw the engine's RNG seed
            # from it, so torch.manual_seed() makes sampled decodes reproducible and successive calls differ
            seed = 0 if top_k == 1 else int(torch.randint(0, 2**62, (1,)).item())
        if B == 2 and _allow_empty is None:
            # Two utterances: measured on MI355X (profiles/r06_small_batch.json), the batched launch chain decodes them at 369 us per AR
            # step -- 41 k tokens/s, LESS than one utterance on the persistent batch-1 launch (128 us per step, 56.7 k) -- so where that
            # launch is available they are decoded one after the other.  Same results as the batched call: utterances never interact,
            # and utterance b draws from the sampling stream of request b (common.h request_seed: the stream of request b under `seed` is
            # the stream of request 0 under seed + b * 0x9E3779B97F4A7C15).  From 3 utterances on the chain is ahead (60.7 k).
            # Since the batched persistent launch (csrc/persist_nb.hip: 2 .. 6 utterances share ONE launch, the weights are streamed once
            # per step for all of them) this is the path only where that launch is not available (fp32 / fp8-weight engines, options).
            eng = self.engine_for(B, max(xl), max(yl))
            if eng.fetch_u32("persist_batch_capable") < B and eng.fetch_u32("persist_capable") == 1:
                outs, tsum = [], dict(prefill_ms=0.0, ar_ms=0.0, nar_ms=0.0, ar_steps=0.0)
                for b in range(B):
                    en_b = None if enroll_x_lens is None else (enroll_x_lens if enroll_x_lens.numel() == 1 else enroll_x_lens[b : b + 1])
                    outs += self.inference_batch(x[b : b + 1, : xl[b]], x_lens[b : b + 1], y[b : b + 1, : yl[b]], [yl[b]], en_b, top_k, temperature,
                                                 (seed + b * 0x9E3779B97F4A7C15) & (2**64 - 1), max_new, _allow_empty=True)
                    for k, v in eng.timings().items():
                        tsum[k] += v
                self.sequential_timings = tsum  # (bench.py: the phase times of the whole call, not of its last utterance)
                return outs
        if _allow_empty is None:
            self.sequential_timings = None
        eng = self.engine_for(B, max(xl), max(yl))
        dev = eng.device
        allow_empty = B > 1 if _allow_empty is None else _allow_empty
        xd, yd = x.to(dev, torch.int64), y.to(dev, torch.int64)[..., : self.num_quantizers]
        try:
            # batch 1 keeps the reference's SyntaxError; in a batch an utterance that hits EOS at step 0 returns 0 frames.
            # (Engine.prefill_generate repeats the decode once on VLE_EBUSY -- the persistent launch could not hold the GPU.)
            _, gl = _engine.prefill_generate(eng, xd, xl, yd, yl, top_k=top_k, temperature=temperature, seed=seed, max_new=max_new, allow_empty=allow_empty)
        except _lib.VleError as err:
            if err.code == _lib.VLE_ENOTOKEN:
                raise SyntaxError("well trained model shouldn't reach here.") from None  # valle.py:1049-1052
            raise
        for b in range(B):
            print(f"VALL-E EOS [{yl[b]} -> {yl[b] + int(self.ar_audio_prepend_bos) + gl[b]}]")  # valle.py:1054
        en = None
        if self.prefix_mode in (2, 4):
            assert enroll_x_lens is not None, "prefix_mode 2/4 needs enroll_x_lens (valle.py:1068-1079)"
            en = [int(v) for v in enroll_x_lens.tolist()]
            if len(en) == 1 and B > 1:
                en = en * B
        codes = eng.nar(en)
        return [codes[b, : gl[b]] for b in range(B)]

    # ---- VALLE.continual (valle/models/valle.py:1139-1238) --------------------------------------
    @torch.no_grad()
    def continual(self, x: torch.Tensor, x_lens: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        assert x.ndim == 2, x.shape
        assert x_lens.ndim == 1, x_lens.shape
        assert y.ndim == 3, y.shape
        assert y.shape[0] == 1, y.shape
        assert torch.all(x_lens > 0)
        assert self.num_quantizers == 8
        T = y.shape[1]
        if not self.fused:
            P = min(int(T * 0.5), 3 * 75)  # valle.py:1173
            dev = self._block_device()
            S = int(x_lens.max())
            return self._nar_blocks(x.to(dev, torch.int64)[:, :S], y.to(dev, torch.int64)[:, :, 0], y.to(dev, torch.int64)[:, :P], P,
                                    1 if self.prefix_mode in (2, 4) else self.prefix_mode, None)
        eng = self.engine_for(1, int(x_lens.max()), T, T)
        dev = eng.device
        codes, gl = eng.continual(x.to(dev, torch.int64), [int(x_lens.max())], y.to(dev, torch.int64), [T])
        return codes[:, : gl[0]]


    # ---- the same decode written on the block modules (constructor combinations outside the fused engine's shape) -----------
    def _block_device(self) -> torch.device:
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("the HIP operators need the model on a ROCm device: call .to('cuda') first (no CPU path)")
        if self.training:
            raise NotImplementedError("decode runs in eval mode: call .eval() first (dropout / BatchNorm statistics)")
        return dev

    def _predict(self, dec: TransformerEncoder, h2d: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
        """nn.Linear(d, V, bias=False) (valle.py:153-155, 248-253) on the GEMM path, fp32 logits."""
        from . import ops

        return ops.linear(h2d.to(dec._tdtype()).contiguous(), dec._w(weight), None, epilogue=ops.EPI_F32)

    def _inference_blocks(self, x, xl, y, yl, enroll_x_lens, top_k, temperature, seed, max_new) -> List[torch.Tensor]:
        """valle.py:993-1137 per utterance, every tensor operation a HIP operator of modules.py / ops.py: like the reference, each
        AR step re-runs the whole [text; audio] sequence through ``ar_decoder`` (no KV cache on this path)."""
        from . import ops

        dev = self._block_device()
        if seed is None:
            seed = 0 if top_k == 1 else int(torch.randint(0, 2**62, (1,)).item())
        bos = int(self.ar_audio_prepend_bos)
        outs = []
        for b in range(x.shape[0]):
            S, P = xl[b], yl[b]
            text = x[b:b + 1, :S].to(dev, torch.int64)
            prompts = y[b:b + 1, :P, : self.num_quantizers].to(dev, torch.int64)
            xe = self.ar_text_position(self.ar_text_prenet(self.ar_text_embedding(text)))            # :994-997
            yy = prompts[..., 0]
            if bos:
                yy = torch.nn.functional.pad(yy, (1, 0), value=NUM_AUDIO_TOKENS + 1)                  # :1006-1007
            n_gen = 0
            while True:
                ye = self.ar_audio_position(self.ar_audio_prenet(self.ar_audio_embedding(yy)))       # :1013-1015
                logits = self._predict(self.ar_decoder, self._ar_last_hidden(xe, ye), self.ar_predict_layer.weight)  # :1016-1039
                # request b's RNG stream at step n_gen: what vle_ar_generate(seed) draws for this utterance
                smp, am = ops.topk_sample(logits, top_k, temperature, seed=_request_seed_base(seed, b), step=n_gen)  # :1040-1042
                smp_i, am_i = int(smp[0]), int(am[0])
                stop = am_i == NUM_AUDIO_TOKENS or smp_i == NUM_AUDIO_TOKENS or (yy.shape[1] - P) > S * 16  # :1044-1048
                if max_new and n_gen >= max_new:
                    stop = True
                if stop:
                    if P == yy.shape[1]:
                        if x.shape[0] == 1:
                            raise SyntaxError("well trained model shouldn't reach here.")               # :1049-1052
                    print(f"{self._eos_name} EOS [{P} -> {yy.shape[1]}]")                              # :1054
                    break
                yy = torch.cat([yy, smp.view(1, 1)], dim=1)                                            # :1057
                n_gen += 1
            y0 = yy[:, bos:]
            en = None
            if self.prefix_mode in (2, 4):
                assert enroll_x_lens is not None, "prefix_mode 2/4 needs enroll_x_lens (valle.py:1068-1079)"
                el = [int(v) for v in enroll_x_lens.tolist()]
                en = el[b] if len(el) > 1 else el[0]
            outs.append(self._nar_blocks(text, y0, prompts, P, self.prefix_mode, en)[0])
        return outs

    def _ar_last_hidden(self, xe: torch.Tensor, ye: torch.Tensor) -> torch.Tensor:
        """Decoder output of the LAST audio position, (1, d): [text; audio] through the decoder-only stack under the prefix-LM mask
        (valle.py:1016-1038)."""
        S, T = xe.shape[1], ye.shape[1]
        i = torch.arange(S + T, device=xe.device)
        allowed = i[None, :] < torch.maximum(i[:, None] + 1, torch.tensor(S, device=xe.device))       # :1018-1033
        h, _ = self.ar_decoder((torch.cat([xe, ye], dim=1), None), mask=~allowed)                     # :1035-1038
        return h[0, -1:]

    def _nar_hidden(self, xe: torch.Tensor, ye: torch.Tensor, stage_weight: torch.Tensor, P: int) -> torch.Tensor:
        """Decoder outputs of the generated frames, (G, d) (valle.py:1123-1128)."""
        h, _ = self.nar_decoder((torch.cat([xe, ye], dim=1), stage_weight))
        return h[0, xe.shape[1] + P:]

    def _fwd_ar_hidden(self, xe: torch.Tensor, ye: torch.Tensor, key_padding_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Teacher-forced AR pass over a batch: decoder outputs of the audio positions, (N, Ta, d) (valle.py:833-872); a padded
        batch passes its (N, S + Ta) padding mask (the reference merges it into the attention mask, :846-856)."""
        S, T = xe.shape[1], ye.shape[1]
        i = torch.arange(S + T, device=xe.device)
        allowed = i[None, :] < torch.maximum(i[:, None] + 1, torch.tensor(S, device=xe.device))       # prefix-LM mask
        h, _ = self.ar_decoder((torch.cat([xe, ye], dim=1), None), mask=~allowed, src_key_padding_mask=key_padding_mask)
        return h[:, S:]

    def _fwd_nar_hidden(self, xe: torch.Tensor, ye: torch.Tensor, stage_weight: torch.Tensor,
                        key_padding_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Teacher-forced NAR pass: decoder outputs of the audio positions, (N, T, d) (valle.py:922-926)."""
        h, _ = self.nar_decoder((torch.cat([xe, ye], dim=1), stage_weight), src_key_padding_mask=key_padding_mask)
        return h[:, xe.shape[1]:]

    def _nar_blocks(self, text, y0, prompts, P: int, prefix_mode: int, enrolled_len) -> torch.Tensor:
        """The seven NAR stages, valle.py:1059-1137 (and continual()'s :1176-1238): text (1, S) ids, y0 (1, P + G) first-codebook
        stream, prompts (1, P, Q) -> codes (1, G, Q)."""
        from . import ops

        Q = self.num_quantizers
        codes = [y0[:, P:]]                                                                           # :1059
        if Q == 1 or y0.shape[1] == P:  # (an utterance of a batch that stopped at its first step: no frames)
            return torch.stack(codes * (1 if Q == 1 else Q), dim=-1)
        if prefix_mode in (2, 4):                                                                      # :1068-1079
            text = torch.cat([text[:, :1], text[:, enrolled_len - 1:]], dim=1)
        S = text.shape[1]
        xe = self.nar_text_position(self.nar_text_prenet(self.nar_text_embedding(text)))              # :1081-1083
        y_emb = self.nar_audio_embeddings[0](y0).clone()                                              # :1064-1066
        if prefix_mode != 0:
            for j in range(1, Q):                                                                      # :1110-1113
                self.nar_audio_embeddings[j].add_to(y_emb[0, :P], prompts[0, :, j])
        for i in range(Q - 1):                                                                         # :1085 / :1115
            ye = self.nar_audio_position(self.nar_audio_prenet(y_emb))                                 # :1121-1122
            logits = self._predict(self.nar_decoder, self._nar_hidden(xe, ye, self.nar_stage_embeddings[i].weight, P),
                                   self.nar_predict_layers[i].weight)                               # :1123-1128
            samples = ops.topk_sample(logits, 1)[1][None]                                              # arg-max, :1130
            codes.append(samples)
            if i < Q - 2:                                                                              # :1133 / :1103
                if prefix_mode == 0:
                    self.nar_audio_embeddings[i + 1].add_to(y_emb[0, :P], prompts[0, :, i + 1])        # :1104-1107
                self.nar_audio_embeddings[i + 1].add_to(y_emb[0, P:], samples[0])                      # :1108 / :1134
        return torch.stack(codes, dim=-1)                                                              # :1136-1137

    # ---- VALLE.forward (valle/models/valle.py:762-959), teacher-forced, eval mode ----------------------------
    @torch.no_grad()
    def forward(self, x: torch.Tensor, x_lens: torch.Tensor, y, y_lens, reduction: str = "sum",
                train_stage: int = 0, *, nar_stage: Optional[int] = None, prefix_len: Optional[int] = None,
                prompt_starts: Optional[Sequence[int]] = None, **kwargs):
        """The reference's teacher-forced pass as a SCORING function (validation loss / Top10Accuracy of given codes):
        returns ``((x_emb, codes), total_loss, metrics)`` like valle.py:959, computed by the HIP block modules (AR: one
        prefix-LM pass over [text; y]; NAR: one unmasked pass at stage ``nar_stage``) and ``vle_op_cross_entropy``.

        Scope: eval mode (no dropout, no gradients), every prefix_mode (0 / 1 / 2; 4 with ``y`` / ``y_lens`` as
        ``PromptedFeatures``, :792-798), ``reduction="sum"``; padded batches (the collater's shapes) for VALL-E since round 3: padded frames blanked,
        padded keys masked, and -- like the reference, whose AR loss has no ignore_index (:875) -- the padded rows' EOS targets
        summed into the AR loss (VALL-F: unpadded batches).  The reference's random draws are keyword arguments; left None they are drawn the way
        the reference draws them: ``nar_stage`` from ``self.rng`` (random.Random(0) at construction, :165, :891-895),
        prefix_mode 1's ``prefix_len`` from torch's global generator (:348-350), prefix_mode 2's per-utterance segment starts
        ``prompt_starts`` from ``self.rng.randint`` (:368-369, after the ``nar_stage`` draw, like the reference)."""
        from . import ops

        assert x.ndim == 2, x.shape
        assert x_lens.ndim == 1, x_lens.shape
        y_prompts_codes = None
        if isinstance(y, PromptedFeatures):                                          # :792-798
            y_prompts_codes, y = y.data
            prompts_len, y_lens = y_lens.data
            assert prompts_len.min() == prompts_len.max()
            assert self.prefix_mode == 4
            y_prompts_codes = y_prompts_codes.type(torch.int64)
        assert y.ndim == 3, y.shape
        assert y_lens.ndim == 1, y_lens.shape
        if self.training:
            raise NotImplementedError("forward() is a scoring pass: call .eval() first (training is outside the decode path)")
        if reduction != "sum":
            raise NotImplementedError("only reduction='sum' (the trainer's, valle/bin/trainer.py) is implemented")
        if self.prefix_mode == 4 and y_prompts_codes is None and train_stage in (0, 2) and self.num_quantizers > 1:
            raise ValueError("prefix_mode 4 takes y / y_lens as PromptedFeatures (prompts, features), like the reference (valle.py:792-798)")
        N, S = x.shape
        T = y.shape[1]
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("the HIP operators need the model on a ROCm device: call .to('cuda') first (no CPU path)")
        xl, yl = [int(v) for v in x_lens], [int(v) for v in y_lens]
        padded = any(v != S for v in xl) or any(v != T for v in yl)
        x_mask = y_mask = None
        if padded:
            # the collater's shapes (valle/data/collation.py): every utterance padded to the longest.  Padded frames are blanked
            # (:811), padded first-codebook inputs / targets are EOS (pad_y_eos :322-333), padded keys are masked (:846-856,
            # :908-926), and the AR loss still sums over the padded rows (:875 has no ignore_index), like the reference's.
            if type(self)._decoder_factory is not _decoder:
                raise NotImplementedError("forward(): VALL-F scores unpadded batches only")
            if max(xl) != S or max(yl) != T or min(xl) < 1 or min(yl) < 1:
                raise ValueError("padded batch: x / y must be padded to their longest utterance (make_pad_mask sizes the masks to it)")
            x_mask = torch.arange(S, device=dev)[None, :] >= torch.tensor(xl, device=dev)[:, None]      # make_pad_mask, :805-806
            y_mask = torch.arange(T, device=dev)[None, :] >= torch.tensor(yl, device=dev)[:, None]
        x, codes = x.to(dev, torch.int64), y.to(dev, torch.int64)
        if padded:
            codes = codes * (~y_mask)[..., None].to(torch.int64)                     # :811
        ymin = min(yl)
        bos = int(self.ar_audio_prepend_bos)
        tdt = self.ar_decoder._tdtype()

        def predict(h2d, weight):  # nn.Linear(d, V, bias=False) on the MFMA GEMM path
            return ops.linear(h2d.to(tdt).contiguous(), self.ar_decoder._w(weight), None, epilogue=ops.EPI_F32)

        total_loss = torch.zeros((), device=dev)
        metrics = {}
        total_length = float(sum(yl))                                                 # y_lens.sum(), :878, :935
        F_pad = torch.nn.functional.pad
        ym_int = y_mask.to(torch.int64) if padded else torch.zeros(N, T, dtype=torch.int64, device=dev)
        t_all = F_pad(codes[..., 0], (0, 1), value=0) + NUM_AUDIO_TOKENS * F_pad(ym_int, (0, 1), value=1)  # pad_y_eos, valle.py:322-333
        if bos:
            inputs, targets = F_pad(t_all[:, :-1], (1, 0), value=NUM_AUDIO_TOKENS + 1), t_all
        else:
            inputs, targets = t_all[:, :-1], t_all[:, 1:]
        if train_stage in (0, 1):
            Ta = inputs.shape[1]
            xe = self.ar_text_position(self.ar_text_prenet(self.ar_text_embedding(x)))            # :827-829
            ye = self.ar_audio_position(self.ar_audio_prenet(self.ar_audio_embedding(inputs)))     # :861-863
            ar_kpm = None
            if padded:
                ar_kpm = torch.cat([x_mask, F_pad(y_mask, (1, 0), value=False) if bos else y_mask], dim=1)  # :820-826
            logits = predict(self._fwd_ar_hidden(xe, ye, ar_kpm).reshape(N * Ta, -1), self.ar_predict_layer.weight)  # :833-873
            loss_rows, hit = ops.cross_entropy_rows(logits, targets.reshape(-1), ignore_index=-100, topk=10)
            total_loss = total_loss + loss_rows.sum()                                 # :875 (no ignore_index)
            kept = targets.reshape(-1) != NUM_AUDIO_TOKENS                           # the metric ignores EOS targets (:157-163)
            acc = (hit[kept] == 1).sum().float() / kept.sum().clamp_min(1).float()
            metrics["ArTop10Accuracy"] = acc * total_length                           # :877-879
        if self.num_quantizers == 1:
            return ((x, codes), total_loss, metrics)
        x_emb = x
        if train_stage in (0, 2):
            if nar_stage is None:
                nq = self.num_quantizers - 1
                nar_stage = self.rng.choices(list(range(1, self.num_quantizers)), weights=[1.0 / nq] * nq, k=1)[0]  # :891-895
            assert 1 <= nar_stage < self.num_quantizers
            P = 0
            if self.prefix_mode == 1:
                if prefix_len is None:
                    int_low = int(0.25 * ymin)
                    prefix_len = min(int(torch.randint(int_low, int_low * 2, size=()).item()), 225)  # :348-350
                P = int(prefix_len)
            xe = self.nar_text_position(self.nar_text_prenet(self.nar_text_embedding(x)))         # :897-899
            x_emb = xe
            y_in = inputs[:, 1:] if bos else inputs                                  # :886-887: first-codebook stream, EOS where padded
            nar_tgt = codes[..., nar_stage] + NUM_AUDIO_TOKENS * ym_int              # :906 (padded rows = ignore_index)
            if self.prefix_mode in (2, 4):
                # _prepare_prompts :362-389: the prompt is a separate segment in front of the WHOLE utterance.  prefix_mode 2 cuts it
                # out of the utterance itself (one self.rng.randint per utterance) and blanks that stretch of the target codebook
                # IN PLACE (the returned codes carry the blanks, like the reference's); prefix_mode 4 gets it from the caller.
                if self.prefix_mode == 2:
                    P = min(225, int(0.25 * ymin))                                   # :364
                    if prompt_starts is None:
                        prompt_starts = [self.rng.randint(0, yl[n] - P) for n in range(N)]  # :368-369
                    assert len(prompt_starts) == N and all(0 <= int(v) <= yl[n] - P for n, v in enumerate(prompt_starts))
                    codes = codes.clone()  # the reference blanks its own copy (:811), never the caller's tensor
                    prompts = torch.stack([codes[n, int(st): int(st) + P].clone() for n, st in enumerate(prompt_starts)])
                    for n, st in enumerate(prompt_starts):
                        codes[n, int(st): int(st) + P, nar_stage] = NUM_AUDIO_TOKENS  # :370-373
                else:
                    prompts = y_prompts_codes.to(dev)
                    assert prompts.shape[0] == N and prompts.shape[2] == self.num_quantizers
                    P = int(prompts.shape[1])                                        # :377
                if self.prefix_mode == 2:
                    nar_tgt = codes[..., nar_stage] + NUM_AUDIO_TOKENS * ym_int      # after the blanking
                y_pr = self.nar_audio_embeddings[0](prompts[..., 0])
                y_full = self.nar_audio_embeddings[0](y_in)
                for j in range(1, self.num_quantizers):
                    self.nar_audio_embeddings[j].add_to(y_pr, prompts[..., j])
                    if j < nar_stage:
                        self.nar_audio_embeddings[j].add_to(y_full, codes[..., j])
                y_emb = torch.cat([y_pr, y_full], dim=1)                            # :389
                targets = nar_tgt.reshape(-1)                                        # :906 (the blanked stretch / padding = ignore_index)
                ye = self.nar_audio_position(self.nar_audio_prenet(y_emb))
                kpm = torch.cat([x_mask, F_pad(y_mask, (P, 0), value=False)], dim=1) if padded else None  # :908-915
                h = self._fwd_nar_hidden(xe, ye, self.nar_stage_embeddings[nar_stage - 1].weight, kpm)
                logits = predict(h[:, P:].reshape(N * T, -1), self.nar_predict_layers[nar_stage - 1].weight)  # :927, VALLF :531-533
                loss_rows, hit = ops.cross_entropy_rows(logits, targets, ignore_index=NUM_AUDIO_TOKENS, topk=10)
                if self.prefix_mode == 4:
                    P = 0                                                            # :929-930: also resets the length correction
                total_loss = total_loss + loss_rows.sum() * (total_length / (total_length - P * N))  # :936-943
                kept = hit >= 0
                metrics["NarTop10Accuracy"] = (hit == 1).sum().float() / kept.sum().clamp_min(1).float() * total_length
                if train_stage == 0:
                    total_loss = total_loss / 2.0
                return ((x_emb, codes), total_loss, metrics)
            y_emb = self.nar_audio_embeddings[0](y_in)                               # _prepare_prompts :335-393
            if self.prefix_mode == 0:
                for j in range(1, nar_stage):
                    self.nar_audio_embeddings[j].add_to(y_emb, codes[..., j])
            else:
                for j in range(1, self.num_quantizers):
                    for n in range(N):  # row ranges of one utterance are contiguous
                        self.nar_audio_embeddings[j].add_to(y_emb[n, :P], codes[n, :P, j])
                        if j < nar_stage:
                            self.nar_audio_embeddings[j].add_to(y_emb[n, P:], codes[n, P:, j])
            targets = nar_tgt[:, P:].reshape(-1)                                     # :906, :916-917
            ye = self.nar_audio_position(self.nar_audio_prenet(y_emb))                             # :919-920
            kpm = torch.cat([x_mask, y_mask], dim=1) if padded else None             # :816
            h = self._fwd_nar_hidden(xe, ye, self.nar_stage_embeddings[nar_stage - 1].weight, kpm)            # :922-926
            logits = predict(h[:, P:].reshape(N * (T - P), -1), self.nar_predict_layers[nar_stage - 1].weight)   # :927-932
            loss_rows, hit = ops.cross_entropy_rows(logits, targets, ignore_index=NUM_AUDIO_TOKENS, topk=10)
            total_loss = total_loss + loss_rows.sum() * (total_length / (total_length - P * N))  # :936-943
            kept = hit >= 0
            metrics["NarTop10Accuracy"] = (hit == 1).sum().float() / kept.sum().clamp_min(1).float() * total_length  # :945-956
        if train_stage == 0:
            total_loss = total_loss / 2.0                                             # :958
        return ((x_emb, codes), total_loss, metrics)


class VALLF(VALLE):
    """VALL-F (valle/models/valle.py:50-710): the same embeddings, prenets, predict layers and NAR scheme as VALL-E, but the
    decoders are ``nn.TransformerDecoder`` stacks -- the text is their cross-attention MEMORY, the audio stream their (causal,
    for AR) target.  Decoded by the HIP block modules in the reference's loop (one full pass per AR step, :613-651); the
    state-dict keys are the reference's (``...multihead_attn...``, ``norm3``).  Not the production model: there is no fused
    engine path and ``continual()`` does not exist in the reference's VALLF; ``forward()`` (the teacher-forced scoring pass,
    valle.py:395-564) is VALLE's with the two decoder passes replaced (same scope: eval mode, unpadded batches, prefix_mode 0 / 1)."""

    _decoder_factory = staticmethod(_cross_decoder)
    _eos_name = "VALL-F"

    def _ar_last_hidden(self, xe: torch.Tensor, ye: torch.Tensor) -> torch.Tensor:
        T = ye.shape[1]
        tgt_mask = torch.triu(torch.ones(T, T, device=ye.device, dtype=torch.bool), diagonal=1)        # valle.py:619-624
        h, _ = self.ar_decoder((ye, None), xe, tgt_mask=tgt_mask, memory_mask=None)                    # :626-632
        return h[0, -1:]

    def _nar_hidden(self, xe: torch.Tensor, ye: torch.Tensor, stage_weight: torch.Tensor, P: int) -> torch.Tensor:
        h, _ = self.nar_decoder((ye, stage_weight), xe, tgt_mask=None, memory_mask=None)               # :691-697
        return h[0, P:]                                                                               # :698

    def _fwd_ar_hidden(self, xe: torch.Tensor, ye: torch.Tensor, key_padding_mask=None) -> torch.Tensor:
        assert key_padding_mask is None  # forward() rejects padded batches for VALL-F before it gets here
        T = ye.shape[1]
        tgt_mask = torch.triu(torch.ones(T, T, device=ye.device, dtype=torch.bool), diagonal=1)        # valle.py:474-480
        h, _ = self.ar_decoder((ye, None), xe, tgt_mask=tgt_mask, memory_mask=None)                    # :481-488
        return h

    def _fwd_nar_hidden(self, xe: torch.Tensor, ye: torch.Tensor, stage_weight: torch.Tensor, key_padding_mask=None) -> torch.Tensor:
        assert key_padding_mask is None
        h, _ = self.nar_decoder((ye, stage_weight), xe, tgt_mask=None, memory_mask=None)               # :537-544
        return h

    def continual(self, *a, **k):
        raise NotImplementedError("VALLF has no continual() (valle/models/valle.py: it is defined on VALLE only)")


# ---- valle/models/__init__.py surface ---------------------------------------------------------------
def _str2bool(v):
    return str(v).lower() in ("1", "true", "yes", "y", "t")


def add_model_arguments(parser: argparse.ArgumentParser):
    """Same flags as valle/models/__init__.py:18-95, plus the engine's own."""
    parser.add_argument("--model-name", type=str, default="VALL-E")
    parser.add_argument("--decoder-dim", type=int, default=1024)
    parser.add_argument("--nhead", type=int, default=16)
    parser.add_argument("--num-decoder-layers", type=int, default=12)
    parser.add_argument("--scale-factor", type=float, default=1.0)
    parser.add_argument("--norm-first", type=_str2bool, default=True)
    parser.add_argument("--add-prenet", type=_str2bool, default=False)
    parser.add_argument("--prefix-mode", type=int, default=0)
    parser.add_argument("--share-embedding", type=_str2bool, default=True)
    parser.add_argument("--prepend-bos", type=_str2bool, default=False)
    parser.add_argument("--num-quantizers", type=int, default=8)
    parser.add_argument("--scaling-xformers", type=_str2bool, default=False)
    parser.add_argument("--engine-dtype", type=str, default="fp32", help="HIP engine arithmetic: fp32 (token-exact), bf16, fp8w (bf16 on fp8 e4m3 weights), or fp8 (fp8w + fp8 activations on the fp8 MFMA)")


def get_model(params) -> nn.Module:
    """valle/models/__init__.py:98-136 for --model-name vall-e|valle (:112-124) and vall-f|vallf (:99-111); the debug
    Transformer-TTS (:125-134) is outside this package's scope."""
    name = str(params.model_name).lower()
    if name not in ("vall-e", "valle", "vall-f", "vallf"):
        raise NotImplementedError(f"model_name={params.model_name!r}: VALL-E and VALL-F are implemented (the debug Transformer-TTS is not)")
    get = params.get if hasattr(params, "get") else lambda k, dflt=None: getattr(params, k, dflt)
    return (VALLF if name in ("vall-f", "vallf") else VALLE)(
        params.decoder_dim,
        params.nhead,
        params.num_decoder_layers,
        norm_first=params.norm_first,
        add_prenet=params.add_prenet,
        prefix_mode=params.prefix_mode,
        share_embedding=params.share_embedding,
        nar_scale_factor=params.scale_factor,
        prepend_bos=params.prepend_bos,
        num_quantizers=params.num_quantizers,
        engine_dtype=get("engine_dtype", "fp32") or "fp32",
    )
