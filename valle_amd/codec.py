"""EnCodec 24 kHz DECODER on the HIP engine: codes -> waveform, the step right after ``VALLE.inference()``.

Drop-in for the decode half of the reference's ``AudioTokenizer`` (valle/data/tokenizer.py:219-242):
``audio_tokenizer.decode([(codes.transpose(2, 1), None)])`` (valle/bin/infer.py:261-263) returns a float tensor
(B, 1, samples) at 24 kHz.  The arithmetic is the third-party ``encodec`` package's (``EncodecModel.encodec_model_24khz()``,
6 kbps = 8 codebooks); it is not installed here and its weights cannot be fetched, so this module takes the decoder's
state dict from the caller (``EncodecModel.encodec_model_24khz().state_dict()`` where the package exists) and runs the
published architecture in ``libvalle_engine.so`` (valle_amd/csrc/codec.hip).  Parity with real weights is UNPINNED; the
implementation is checked against oracle/encodec_oracle.py on synthetic weights.  No PyTorch / CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib

SAMPLE_RATE = 24000
HOP = 320  # samples per code frame (75 Hz)
BINS = 1024


def _ccheck(lib, code: int, h):
    if code == _lib.VLE_OK:
        return
    msg = lib.vle_codec_last_error(h)
    raise _lib.VleError(code, msg.decode() if msg else "?")


class EncodecDecoder:
    """Holds a vle_codec*: ``decode_codes`` for one (T, Q) code matrix, ``decode`` with the reference's frame-list signature."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], device="cuda:0", num_quantizers: int = 8):
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the HIP codec needs a ROCm device (no CPU path)")
        self.num_quantizers = num_quantizers
        h = C.c_void_p()
        code = self.lib.vle_codec_create(self.device.index or 0, num_quantizers, C.byref(h))
        if code != _lib.VLE_OK:
            _lib.check(code, None)
        self.h = h
        used = 0
        for key, t in state_dict.items():
            if not (key.startswith("decoder.") or key.startswith("quantizer.")) or not torch.is_floating_point(t):
                continue  # encoder.*, EMA buffers (cluster_size, inited, embed_avg are harmless but unused)
            if key.endswith("cluster_size") or key.endswith("embed_avg") or key.endswith("inited"):
                continue
            t = t.detach().to("cpu", torch.float32).contiguous()
            if t.dim() < 1 or t.dim() > 3:
                continue
            shape = (C.c_int64 * t.dim())(*t.shape)
            _ccheck(self.lib, self.lib.vle_codec_load_tensor(self.h, key.encode(), C.c_void_p(t.data_ptr()), shape, t.dim()), self.h)
            used += 1
        _ccheck(self.lib, self.lib.vle_codec_finalize(self.h), self.h)
        self.sample_rate = SAMPLE_RATE
        self.channels = 1

    def close(self):
        if getattr(self, "h", None):
            self.lib.vle_codec_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @torch.no_grad()
    def decode_codes(self, codes: torch.Tensor) -> torch.Tensor:
        """codes int64 (T, Q) -- the layout ``VALLE.inference()`` returns per utterance -- -> wav fp32 (320 T,) on the device."""
        assert codes.dim() == 2 and codes.shape[1] == self.num_quantizers, codes.shape
        codes = codes.to(self.device, torch.int64).contiguous()
        if codes.numel():
            lo, hi = torch.aminmax(codes)
            if int(lo) < 0 or int(hi) >= BINS:  # the reference's F.embedding raises IndexError
                raise IndexError(f"code out of range for a {BINS}-entry codebook: min {int(lo)}, max {int(hi)}")
        T = codes.shape[0]
        wav = torch.empty(T * HOP, dtype=torch.float32, device=self.device)
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _ccheck(self.lib, self.lib.vle_codec_decode(self.h, st, C.c_void_p(codes.data_ptr()), T, C.c_void_p(wav.data_ptr())), self.h)
        return wav

    @torch.no_grad()
    def decode(self, frames: Sequence[Tuple[torch.Tensor, Optional[torch.Tensor]]]) -> torch.Tensor:
        """``EncodecModel.decode`` / ``AudioTokenizer.decode`` signature: a list with ONE (codes (B, Q, T), scale None) frame
        (the 24 kHz model does not segment or rescale) -> (B, 1, 320 T)."""
        assert len(frames) == 1, "the 24 kHz model decodes a single frame (no segmenting)"
        codes, scale = frames[0]
        assert scale is None, "encodec_model_24khz has normalize=False: no scale"
        assert codes.dim() == 3, codes.shape
        outs: List[torch.Tensor] = [self.decode_codes(codes[b].transpose(0, 1)) for b in range(codes.shape[0])]
        return torch.stack(outs)[:, None, :]
