"""ctypes binding of libvalle_engine.so (the C ABI declared in include/valle_engine.h).

The library is the product: there is NO Python/PyTorch fallback.  If it is missing or does
not export the expected symbols, loading fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VLE_LIB: another build of the same ABI (the `make asan` library); the default is the in-tree library
LIB_PATH = os.environ.get("VLE_LIB") or os.path.join(_HERE, "libvalle_engine.so")

VLE_OK = 0
VLE_EINVAL, VLE_ESTATE, VLE_EHIP, VLE_EKEY, VLE_ENOTOKEN, VLE_EINDEX, VLE_EBUSY = -1, -2, -3, -4, -5, -6, -7
DTYPE_F32, DTYPE_BF16, DTYPE_FP8W, DTYPE_FP8 = 0, 1, 2, 3


class VleConfig(C.Structure):
    _fields_ = [
        ("d_model", C.c_int32), ("nhead", C.c_int32), ("num_layers", C.c_int32), ("num_quantizers", C.c_int32),
        ("prefix_mode", C.c_int32), ("prepend_bos", C.c_int32), ("norm_first", C.c_int32), ("add_prenet", C.c_int32),
        ("dtype_mode", C.c_int32), ("max_batch", C.c_int32), ("max_text", C.c_int32), ("max_prompt", C.c_int32),
        ("max_gen", C.c_int32), ("device", C.c_int32), ("use_graph", C.c_int32), ("steps_per_graph", C.c_int32),
        ("reserved", C.c_int32 * 8),
    ]


_P = C.c_void_p
_I32P = C.POINTER(C.c_int32)

# name -> (restype, argtypes); must list every symbol include/valle_engine.h declares
SIGNATURES = {
    "vle_create": (C.c_int, [C.POINTER(VleConfig), C.POINTER(_P)]),
    "vle_destroy": (None, [_P]),
    "vle_last_error": (C.c_char_p, [_P]),
    "vle_load_tensor": (C.c_int, [_P, C.c_char_p, _P, C.POINTER(C.c_int64), C.c_int]),
    "vle_finalize_weights": (C.c_int, [_P]),
    "vle_ar_prefill": (C.c_int, [_P, _P, _P, C.c_int64, _I32P, _P, C.c_int64, _I32P, C.c_int32]),
    "vle_ar_generate": (C.c_int, [_P, _P, C.c_int32, C.c_float, C.c_uint64, C.c_int32, _P, C.c_int64, _I32P, _P, C.c_int64, _I32P]),
    "vle_nar_decode": (C.c_int, [_P, _P, _I32P, _P, C.c_int64]),
    "vle_reserve": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int64]),
    "vle_nar_force": (C.c_int, [_P, _P, C.c_int64]),
    "vle_nar_continual": (C.c_int, [_P, _P, _P, C.c_int64, _I32P, _P, C.c_int64, _I32P, C.c_int32, _P, C.c_int64, _I32P]),
    "vle_slots_begin": (C.c_int, [_P, _P]),
    "vle_slots_prefill": (C.c_int, [_P, _P, C.c_int32, _I32P, _P, C.c_int64, _I32P, _P, C.c_int64, _I32P, C.c_int32, C.c_float, C.c_uint64]),
    "vle_slots_step": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_float, C.c_uint64, _I32P, _I32P]),
    "vle_slots_harvest": (C.c_int, [_P, _P, C.c_int32, _I32P, _I32P, _P, C.c_int64]),
    "vle_set_option": (C.c_int, [_P, C.c_char_p, C.c_int64]),
    "vle_debug_fetch": (C.c_int64, [_P, C.c_char_p, _P, C.c_size_t]),
    "vle_last_timings": (C.c_int, [_P, C.POINTER(C.c_double)]),
    "vle_debug_guard_alloc": (C.c_int, [C.c_int32, C.c_size_t, C.c_int32, C.POINTER(_P)]),
    "vle_ar_step_bytes": (C.c_int64, [_P, C.c_int32, C.c_int64]),
    "vle_op_layernorm": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, C.c_int64, C.c_int32]),
    "vle_op_linear": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int]),
    "vle_op_linear_ws": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int, _P, C.c_int32]),
    "vle_op_linear_workspace_bytes": (C.c_int64, []),
    "vle_op_linear_ln_producer": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, C.c_int64, C.c_int32, C.c_int32]),
    "vle_op_linear_ln_consumer": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32]),
    "vle_op_tune": (C.c_int, [C.c_char_p, C.c_int64]),
    "vle_quantize_fp8w": (C.c_int, [_P, C.c_int64, C.c_int64, _P, _P, _P]),
    "vle_op_linear_skinny_fp8w": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int]),
    "vle_op_linear_fp8w": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int, _P, C.c_int32]),
    "vle_op_linear_skinny": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int]),
    "vle_op_attention": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int]),
    "vle_op_attn_step1": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "vle_op_decode_attention": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "vle_op_attn_out_proj": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "vle_op_token_embedding": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int32]),
    "vle_op_token_embedding_add": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int32]),
    "vle_op_sine_positional": (C.c_int, [_P, _P, _P, _P, C.c_float, _P, C.c_int64, C.c_int32, C.c_int32]),
    "vle_codec_create": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(_P)]),
    "vle_codec_load_tensor": (C.c_int, [_P, C.c_char_p, _P, C.POINTER(C.c_int64), C.c_int]),
    "vle_codec_finalize": (C.c_int, [_P]),
    "vle_codec_decode": (C.c_int, [_P, _P, _P, C.c_int64, _P]),
    "vle_codec_last_error": (C.c_char_p, [_P]),
    "vle_codec_destroy": (None, [_P]),
    "vle_op_quantize_rows_fp8": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int32]),
    "vle_op_linear_fp8": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int]),
    "vle_op_cross_entropy": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32]),
    "vle_op_cross_attention": (C.c_int, [_P, C.c_int, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "vle_op_topk_sample": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_uint64, C.c_uint32, _P, _P]),
    "vle_op_adaln_fold": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int32]),
}

_lib = None


def load():
    """Load the shared library (once).  Raises RuntimeError when it is absent or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP engine is not built. Run `python __graft_entry__.py build` "
            "(hipcc --offload-arch=gfx950). There is no fallback path."
        )
    # torch ships its own libamdhip64.so (same SONAME): import it first so both share one HIP runtime
    import torch  # noqa: F401

    lib = C.CDLL(LIB_PATH, mode=C.RTLD_LOCAL)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            raise RuntimeError(f"{LIB_PATH} does not export {name}")
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


class VleError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"valle_engine error {code}: {msg}")
        self.code = code


def check(code: int, handle=None):
    if code == VLE_OK:
        return
    msg = load().vle_last_error(handle)
    if code == VLE_EINDEX:  # what nn.Embedding raises in the reference (valle/modules/embedding.py:34,44)
        raise IndexError(msg.decode() if msg else "token id out of range")
    raise VleError(code, msg.decode() if msg else "?")


class _GuardedMemory:
    """`bytes` of device memory from vle_debug_guard_alloc, exposed through __cuda_array_interface__ (torch.as_tensor wraps it)."""

    def __init__(self, nbytes: int, device: int, at_start: bool):
        p = _P()
        check(load().vle_debug_guard_alloc(device, nbytes, int(at_start), C.byref(p)))
        self.ptr = p.value
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 2}


def guarded_like(t, at_start: bool = False):
    """Debugging aid: a copy of the CUDA tensor `t` whose storage ends (starts) at the edge of its own virtual-memory mapping, an
    unmapped page behind (in front of) it -- see vle_debug_guard_alloc.  The memory is never freed."""
    import torch

    t = t.contiguous()
    nbytes = max(t.numel() * t.element_size(), 1)
    mem = _GuardedMemory(nbytes, t.device.index or 0, at_start)
    raw = torch.as_tensor(mem, device=t.device)
    out = raw[: t.numel() * t.element_size()].view(t.dtype).view(t.shape) if t.numel() else t.clone()
    out.copy_(t)
    out._vle_guard = mem
    return out
