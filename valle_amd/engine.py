"""Python handle on the C-ABI engine: owns a vle_engine*, moves tensors in and out.

PyTorch is plumbing here (device memory, streams); every arithmetic op of the decode path
runs in libvalle_engine.so.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import torch

from . import _lib

NUM_AUDIO_TOKENS = 1024  # valle/models/macros.py:5


def prefill_generate(eng, text, text_lens, prompts, prompt_lens, **gen):
    """prefill() + generate(**gen) with the ONE retry the persistent batch-1 launch asks of its callers: that launch needs every
    CU of the GPU; when another workload holds some for > 0.1 s a wave gives up, vle_ar_generate ends with VLE_EBUSY and the
    call's state is void.  The decode is repeated from the prefill -- the engine itself keeps its next batch-1 calls on the launch
    chain (same sampling stream; same tokens up to the fp32 re-association of the folded LayerNorm) and re-arms the persistent
    launch after a back-off (2, 4 ... 64 calls): a busy neighbour costs speed for a while, not the request.  Every caller that
    wants a decode rather than the raw ABI answer goes through here (VALLE.inference_batch, bench.py, smoke)."""
    eng.prefill(text, text_lens, prompts, prompt_lens)
    try:
        return eng.generate(**gen)
    except _lib.VleError as err:
        if err.code != _lib.VLE_EBUSY:
            raise
        import sys

        print("valle_amd: the persistent AR launch could not hold the whole GPU; this decode is repeated on the launch chain "
              f"(fallback #{eng.fetch_u32('persist_fallbacks')}, persistent launch re-armed after {eng.fetch_u32('persist_backoff')} calls)",
              file=sys.stderr)
        eng.prefill(text, text_lens, prompts, prompt_lens)
        return eng.generate(**gen)


@dataclass
class EngineConfig:
    d_model: int
    nhead: int
    num_layers: int
    num_quantizers: int = 8
    prefix_mode: int = 0
    prepend_bos: bool = False
    dtype: str = "bf16"  # "fp32" (token-exact) | "bf16" | "fp8w" (bf16 arithmetic on fp8-representable weights)
    #                      | "fp8" (fp8w + fp8 activations on CDNA4's block-scaled fp8 MFMA in the prefill / NAR passes)
    max_batch: int = 1
    max_text: int = 64
    max_prompt: int = 225
    max_gen: int = 0  # 0 => 16 * max_text + 1
    device: int = 0
    use_graph: bool = True
    steps_per_graph: int = 8

    def max_gen_eff(self) -> int:
        return self.max_gen if self.max_gen > 0 else 16 * self.max_text + 1

    def max_pos(self) -> int:
        return max(self.max_text, self.max_prompt + 1 + self.max_gen_eff()) + 1


def sine_pe(max_pos: int, d: int) -> torch.Tensor:
    """The table SinePositionalEmbedding.extend_pe builds (valle/modules/embedding.py:75-91),
    by the same torch fp32 ops so the engine adds bit-identical values."""
    pe = torch.zeros(max_pos, d)
    position = torch.arange(0, max_pos, dtype=torch.float32).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def _i32(vals: Sequence[int]):
    return (C.c_int32 * len(vals))(*[int(v) for v in vals])


def _stream_ptr(device: torch.device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class Engine:
    def __init__(self, cfg: EngineConfig):
        self.cfg = cfg
        self.lib = _lib.load()
        c = _lib.VleConfig()
        c.d_model, c.nhead, c.num_layers, c.num_quantizers = cfg.d_model, cfg.nhead, cfg.num_layers, cfg.num_quantizers
        c.prefix_mode, c.prepend_bos, c.norm_first, c.add_prenet = cfg.prefix_mode, int(cfg.prepend_bos), 1, 0
        c.dtype_mode = {"fp32": _lib.DTYPE_F32, "f32": _lib.DTYPE_F32, "bf16": _lib.DTYPE_BF16, "fp8w": _lib.DTYPE_FP8W, "fp8": _lib.DTYPE_FP8}[cfg.dtype]
        c.max_batch, c.max_text, c.max_prompt, c.max_gen = cfg.max_batch, cfg.max_text, cfg.max_prompt, cfg.max_gen
        c.device, c.use_graph, c.steps_per_graph = cfg.device, int(cfg.use_graph), cfg.steps_per_graph
        h = C.c_void_p()
        _lib.check(self.lib.vle_create(C.byref(c), C.byref(h)))
        self.h = h
        self.device = torch.device("cuda", cfg.device)
        self._B = 0
        self._gen_lens: List[int] = []

    def close(self):
        if getattr(self, "h", None):
            self.lib.vle_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- weights ------------------------------------------------------------------------------
    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        for key, t in sd.items():
            t = t.detach().to(device="cpu", dtype=torch.float32).contiguous()
            shape = (C.c_int64 * t.dim())(*t.shape)
            _lib.check(self.lib.vle_load_tensor(self.h, key.encode(), C.c_void_p(t.data_ptr()), shape, t.dim()), self.h)
        pe = sine_pe(self.cfg.max_pos(), self.cfg.d_model).contiguous()
        shape = (C.c_int64 * 2)(*pe.shape)
        _lib.check(self.lib.vle_load_tensor(self.h, b"position.pe", C.c_void_p(pe.data_ptr()), shape, 2), self.h)
        _lib.check(self.lib.vle_finalize_weights(self.h), self.h)

    def reserve(self, max_batch: int, max_text: int, max_prompt: int, max_gen: int = 0):
        """Grow capacities in place (vle_reserve): the weights stay on the device, buffers are re-created."""
        c = self.cfg
        nb, ns, npp = max(c.max_batch, max_batch), max(c.max_text, max_text), max(c.max_prompt, max_prompt)
        ng = max(c.max_gen_eff(), max_gen, 16 * ns + 1)
        new = EngineConfig(**{**c.__dict__, "max_batch": nb, "max_text": ns, "max_prompt": npp, "max_gen": ng})
        pe = sine_pe(new.max_pos(), c.d_model).contiguous()
        _lib.check(self.lib.vle_reserve(self.h, nb, ns, npp, ng, C.c_void_p(pe.data_ptr()), pe.shape[0]), self.h)
        self.cfg = new
        self._B, self._gen_lens = 0, []

    def set_option(self, name: str, value: int):
        _lib.check(self.lib.vle_set_option(self.h, name.encode(), int(value)), self.h)

    # ---- phases -------------------------------------------------------------------------------
    def prefill(self, text: torch.Tensor, text_lens: Sequence[int], prompts: torch.Tensor, prompt_lens: Sequence[int]):
        """text int64 (B,S) and prompts int64 (B,P,Q) on the engine's device."""
        assert text.dtype == torch.int64 and prompts.dtype == torch.int64
        assert text.device == self.device and prompts.device == self.device, "inputs must be on the engine's GPU"
        text, prompts = text.contiguous(), prompts.contiguous()
        B = text.shape[0]
        assert prompts.shape[0] == B and prompts.shape[2] == self.cfg.num_quantizers
        self._keep = (text, prompts)
        _lib.check(
            self.lib.vle_ar_prefill(
                self.h, _stream_ptr(self.device), C.c_void_p(text.data_ptr()), text.shape[1], _i32(text_lens),
                C.c_void_p(prompts.data_ptr()), prompts.shape[1], _i32(prompt_lens), B,
            ),
            self.h,
        )
        self._B = B

    def generate(self, top_k: int = -100, temperature: float = 1.0, seed: int = 0, max_new: int = 0,
                 forced: Optional[torch.Tensor] = None, forced_lens: Optional[Sequence[int]] = None,
                 allow_empty: bool = False):
        """Runs the AR loop; returns (codes0 int64 (B, max_gen) device tensor, gen_lens list)."""
        B = self._B
        G = self.cfg.max_gen_eff()
        codes0 = torch.zeros(B, G, dtype=torch.int64, device=self.device)
        gl = (C.c_int32 * B)()
        f_ptr, f_stride, f_lens = None, 0, None
        if forced is not None:
            forced = forced.to(self.device, torch.int64).contiguous()
            assert forced.dim() == 2 and forced.shape[0] == B
            f_ptr, f_stride = C.c_void_p(forced.data_ptr()), forced.shape[1]
            f_lens = _i32(forced_lens if forced_lens is not None else [forced.shape[1]] * B)
        rc = self.lib.vle_ar_generate(
            self.h, _stream_ptr(self.device), int(top_k), float(temperature), int(seed) & (2**64 - 1), int(max_new),
            f_ptr, f_stride, f_lens, C.c_void_p(codes0.data_ptr()), G, gl,
        )
        self._gen_lens = [int(v) for v in gl]
        if not (allow_empty and rc == _lib.VLE_ENOTOKEN):  # batch: an utterance that stops at step 0 yields 0 frames
            _lib.check(rc, self.h)
        return codes0, self._gen_lens

    def prefill_generate(self, text: torch.Tensor, text_lens: Sequence[int], prompts: torch.Tensor, prompt_lens: Sequence[int], **gen):
        """prefill() + generate(**gen) with the one repeat VLE_EBUSY asks for: the module-level `prefill_generate` on this engine."""
        return prefill_generate(self, text, text_lens, prompts, prompt_lens, **gen)

    def nar(self, enroll_lens: Optional[Sequence[int]] = None, forced: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The 7 NAR stages; ``forced`` int64 (B, >=G, Q) teacher-forces the stage history (parity hook, vle_nar_force)."""
        B, Q = self._B, self.cfg.num_quantizers
        Gmax = max(max(self._gen_lens), 1)
        codes = torch.zeros(B, Gmax, Q, dtype=torch.int64, device=self.device)
        el = _i32(enroll_lens) if enroll_lens is not None else None
        if forced is not None:
            forced = forced.to(self.device, torch.int64).contiguous()
            assert forced.dim() == 3 and forced.shape[0] == B and forced.shape[2] == Q and forced.shape[1] >= max(self._gen_lens)
            _lib.check(self.lib.vle_nar_force(self.h, C.c_void_p(forced.data_ptr()), forced.shape[1]), self.h)
        _lib.check(self.lib.vle_nar_decode(self.h, _stream_ptr(self.device), el, C.c_void_p(codes.data_ptr()), Gmax), self.h)
        return codes

    def continual(self, text: torch.Tensor, text_lens: Sequence[int], y: torch.Tensor, y_lens: Sequence[int]):
        text, y = text.contiguous(), y.contiguous()
        B = text.shape[0]
        Gmax = max(int(t) - min(int(t * 0.5), 225) for t in y_lens)
        Gmax = max(Gmax, 1)
        codes = torch.zeros(B, Gmax, 8, dtype=torch.int64, device=self.device)
        gl = (C.c_int32 * B)()
        _lib.check(
            self.lib.vle_nar_continual(
                self.h, _stream_ptr(self.device), C.c_void_p(text.data_ptr()), text.shape[1], _i32(text_lens),
                C.c_void_p(y.data_ptr()), y.shape[1], _i32(y_lens), B, C.c_void_p(codes.data_ptr()), Gmax, gl,
            ),
            self.h,
        )
        self._B = B
        self._gen_lens = [int(v) for v in gl]
        return codes, self._gen_lens

    # ---- slot API: continuous batching (vle_slots_*) ---------------------------------------------
    def slots_begin(self):
        _lib.check(self.lib.vle_slots_begin(self.h, _stream_ptr(self.device)), self.h)
        self._B = self.cfg.max_batch

    def slots_prefill(self, slots: Sequence[int], text: torch.Tensor, text_lens: Sequence[int], prompts: torch.Tensor,
                      prompt_lens: Sequence[int], top_k: int = 1, temperature: float = 1.0, seed: int = 0):
        """Admit len(slots) utterances into free slots: text int64 (n, S), prompts int64 (n, P, Q) on the engine's device."""
        assert text.dtype == torch.int64 and prompts.dtype == torch.int64 and text.device == self.device and prompts.device == self.device
        text, prompts = text.contiguous(), prompts.contiguous()
        n = len(slots)
        assert text.shape[0] == n and prompts.shape[0] == n and prompts.shape[2] == self.cfg.num_quantizers
        _lib.check(self.lib.vle_slots_prefill(self.h, _stream_ptr(self.device), n, _i32(slots), C.c_void_p(text.data_ptr()), text.shape[1],
                                              _i32(text_lens), C.c_void_p(prompts.data_ptr()), prompts.shape[1], _i32(prompt_lens),
                                              int(top_k), float(temperature), int(seed) & (2**64 - 1)), self.h)

    def slots_step(self, nsteps: int, top_k: int = 1, temperature: float = 1.0, seed: int = 0):
        """Advance every live slot by nsteps AR steps; returns (done flags, generated lengths) per slot."""
        B = self.cfg.max_batch
        done, gl = (C.c_int32 * B)(), (C.c_int32 * B)()
        _lib.check(self.lib.vle_slots_step(self.h, _stream_ptr(self.device), int(nsteps), int(top_k), float(temperature),
                                           int(seed) & (2**64 - 1), done, gl), self.h)
        return [int(v) for v in done], [int(v) for v in gl]

    def slots_harvest(self, slots: Sequence[int], gen_lens: Sequence[int], enroll_lens: Optional[Sequence[int]] = None):
        """NAR stages of the listed finished slots; returns one (G, Q) int64 device tensor per listed slot and frees them."""
        Q = self.cfg.num_quantizers
        Gmax = max(max(gen_lens), 1)
        codes = torch.zeros(self.cfg.max_batch, Gmax, Q, dtype=torch.int64, device=self.device)
        el = _i32(enroll_lens) if enroll_lens is not None else None
        _lib.check(self.lib.vle_slots_harvest(self.h, _stream_ptr(self.device), len(slots), _i32(slots), el, C.c_void_p(codes.data_ptr()), Gmax),
                   self.h)
        return [codes[s, :g] for s, g in zip(slots, gen_lens)]

    # ---- hooks --------------------------------------------------------------------------------
    def timings(self) -> Dict[str, float]:
        out = (C.c_double * 4)()
        _lib.check(self.lib.vle_last_timings(self.h, out), self.h)
        return dict(prefill_ms=out[0], ar_ms=out[1], nar_ms=out[2], ar_steps=out[3])

    def ar_step_bytes(self, B: int, sum_ctx: int) -> int:
        return int(self.lib.vle_ar_step_bytes(self.h, B, sum_ctx))

    def fetch_ar_logits(self) -> torch.Tensor:
        steps = int(self.timings()["ar_steps"]) + 1
        out = torch.empty(steps, self._B, NUM_AUDIO_TOKENS + 1, dtype=torch.float32)
        n = self.lib.vle_debug_fetch(self.h, b"ar_logits", C.c_void_p(out.data_ptr()), out.numel() * 4)
        if n < 0:
            _lib.check(int(n), self.h)
        return out

    def fetch_nar_logits(self, stage: int) -> torch.Tensor:
        rows = sum(self._gen_lens)
        out = torch.empty(rows, NUM_AUDIO_TOKENS, dtype=torch.float32)
        n = self.lib.vle_debug_fetch(self.h, f"nar_logits:{stage}".encode(), C.c_void_p(out.data_ptr()), out.numel() * 4)
        if n < 0:
            _lib.check(int(n), self.h)
        return out

    def kernel_times(self) -> Dict[str, float]:
        """Average microseconds per launch of each AR-step kernel family (after "profile_kernels")."""
        buf = (C.c_double * 16)()
        n = self.lib.vle_debug_fetch(self.h, b"kernel_times", buf, 16 * 8)
        if n < 0:
            _lib.check(int(n), self.h)
        names = ["qkv", "decode_attention", "out_proj", "ffn1", "ffn2", "logits", "sample"]
        return {nm: round(buf[i] * 1e3 / buf[8 + i], 3) for i, nm in enumerate(names) if buf[8 + i] > 0}

    def fetch_ktrace(self) -> torch.Tensor:
        """(32 step slots, 64 kernel slots, 8) int64 wall-clock stamps in 10 ns ticks (option "ktrace"), reduced over the
        waves of each kernel: [first start, last start, first mark1, last mark1, first mark2, last mark2, first end, last end]
        (kernels without phase marks repeat the end stamp there); -1 where nothing was stamped."""
        raw = torch.empty(32, 64, 2048, 4, dtype=torch.int64)
        n = self.lib.vle_debug_fetch(self.h, b"ktrace", C.c_void_p(raw.data_ptr()), raw.numel() * 8)
        if n < 0:
            _lib.check(int(n), self.h)
        valid = raw[..., 0] != -1
        big = torch.iinfo(torch.int64).max
        cols = []
        for k in range(4):
            v = raw[..., k]
            cols += [torch.where(valid, v, big).amin(-1), torch.where(valid, v, -1).amax(-1)]
        out = torch.stack(cols, dim=-1)
        out[~valid.any(-1)] = -1
        self.ktrace_waves = valid.sum(-1)
        return out

    def fetch_u32(self, what: str) -> int:
        """One 32-bit diagnostic word: "qa_spin_fail", "gs_gran_fail", "persist_fail", "persist_active"."""
        v = C.c_uint32(0)
        n = self.lib.vle_debug_fetch(self.h, what.encode(), C.byref(v), 4)
        if n < 0:
            _lib.check(int(n), self.h)
        return int(v.value)

    def fetch_persist_trace(self) -> torch.Tensor:
        """(8 step slots, 256 workgroups, 512) int64 words of the persistent step's timeline (option "persist_trace"; the traced
        instantiations exist for persist_nk = 2 / persist_pf = 3): triplets {wall clock (10 ns ticks) when the workgroup's thread 0
        began to wait for a hand-off, polling passes it took, wall clock when the data was in LDS}, the first triplet = workgroup
        entry; 0 where nothing was stamped."""
        raw = torch.zeros(8, 256, 512, dtype=torch.int64)
        n = self.lib.vle_debug_fetch(self.h, b"persist_trace", C.c_void_p(raw.data_ptr()), raw.numel() * 8)
        if n < 0:
            _lib.check(int(n), self.h)
        return raw

    def fetch_sampled(self) -> torch.Tensor:
        out = torch.empty(self._B, self.cfg.max_gen_eff(), dtype=torch.int64)
        n = self.lib.vle_debug_fetch(self.h, b"ar_sampled", C.c_void_p(out.data_ptr()), out.numel() * 8)
        if n < 0:
            _lib.check(int(n), self.h)
        return out
