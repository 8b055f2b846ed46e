"""ctypes binding of libhhrl_hip.so (declared in include/hhrl_hip.h).

The product path has no CPU fallback: if the shared library is missing or a
call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libhhrl_hip.so")

ACT_NONE, ACT_RELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3

_P = C.c_void_p
_I = C.c_int
_L = C.c_long
_F = C.c_float
_D = C.c_double

ABI_VERSION = 104            # HHRL_ABI_VERSION of include/hhrl_hip.h


class LstmDesc(C.Structure):
    """hhrl_lstm_desc of include/hhrl_hip.h."""
    _fields_ = [("obs", _P), ("ldobs", _I), ("o", _I), ("params", _P), ("grads", _P), ("out", _P), ("ldo", _I),
                ("d_out", _P), ("lddo", _I), ("ws", _P), ("no_backward", _I)]


LSTM_GROUP_MAX = 16


class CopyDesc(C.Structure):
    """hhrl_copy_desc of include/hhrl_hip.h."""
    _fields_ = [("inp", _P), ("ldi", _I), ("out", _P), ("ldo", _I), ("C", _I)]


COPY_GROUP_MAX = 32


class Mlp3Desc(C.Structure):
    """hhrl_mlp3_desc of include/hhrl_hip.h."""
    _fields_ = [("x", _P), ("ldx", _I), ("K0", _I),
                ("w1", _P), ("b1", _P), ("w2", _P), ("b2", _P), ("w3", _P), ("b3", _P),
                ("N3", _I), ("act3", _I), ("h1", _P), ("h2", _P), ("y", _P), ("ldy", _I),
                ("dy", _P), ("lddy", _I), ("dy_const", _F),
                ("gw1", _P), ("gb1", _P), ("gw2", _P), ("gb2", _P), ("gw3", _P), ("gb3", _P),
                ("dx", _P), ("lddx", _I),
                ("smooth_eps", _P), ("y2", _P), ("ldy2", _I), ("forward_dx", _I),
                ("gslab", _P), ("gslab_floats", _L),
                ("mse_q1t", _P), ("mse_q2t", _P), ("mse_r", _P), ("mse_ldr", _I), ("mse_gamma", _F), ("mse_inv_n", _F),
                ("mse_loss", _P), ("mse_target", _P)]


MLP3_GROUP_MAX = 8


class AdamRide(C.Structure):
    """hhrl_adam_ride of include/hhrl_hip.h."""
    _fields_ = [("grads", _P), ("params", _P), ("m", _P), ("v", _P), ("n", _L), ("hyper_dev", _P),
                ("target", _P), ("tau", C.c_double)]


class GCopyDesc(C.Structure):
    """hhrl_gcopy_desc of include/hhrl_hip.h."""
    _fields_ = [("inp", _P), ("ldi", _L), ("out", _P), ("ldo", _I), ("C", _I), ("gather", _I)]


class LinearFwdDesc(C.Structure):
    """hhrl_linear_fwd_desc of include/hhrl_hip.h."""
    _fields_ = [("x", _P), ("ldx", _I), ("w", _P), ("b", _P), ("y", _P), ("ldy", _I),
                ("K", _I), ("N", _I), ("act", _I), ("accumulate", _I),
                ("smooth_eps", _P), ("y2", _P), ("ldy2", _I)]


class LinearBwdDesc(C.Structure):
    """hhrl_linear_bwd_desc of include/hhrl_hip.h."""
    _fields_ = [("x", _P), ("ldx", _I), ("w", _P), ("y", _P), ("ldy", _I), ("dy", _P), ("lddy", _I),
                ("dx", _P), ("lddx", _I), ("dx_accumulate", _I), ("dw", _P), ("db", _P),
                ("K", _I), ("N", _I), ("act", _I)]

# name -> (restype, argtypes); mirrors include/hhrl_hip.h one to one
SIGNATURES = {
    "hhrl_version": (_I, []),
    "hhrl_last_error": (C.c_char_p, []),
    "hhrl_linear_forward": (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "hhrl_linear_backward": (_I, [_P, _I, _P, _P, _I, _P, _I, _P, _I, _I, _P, _P, _I, _I, _I, _I, _P]),
    "hhrl_linear_backward_row": (_I, [_P, _P, _I, _P, _F, _P, _I, _I, _I, _I, _I, _I, _P]),
    "hhrl_linear_backward_cols": (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "hhrl_linear_forward_group": (_I, [_P, _I, _I, _P]),
    "hhrl_linear_backward_group": (_I, [_P, _I, _I, _P]),
    "hhrl_mlp3_supported": (_I, [_I, _I, _I, _I]),
    "hhrl_mlp3_slab_floats": (_L, [_I, _I, _I]),
    "hhrl_mlp3_forward_group": (_I, [_P, _I, _I, _P]),
    "hhrl_mlp3_backward_group": (_I, [_P, _I, _I, _P]),
    "hhrl_mlp3_backward_group_adam": (_I, [_P, _I, _I, _P, _P]),
    "hhrl_sum_pool": (_I, [_P, _I, _P, _I, _I, _I, _I, _P]),
    "hhrl_broadcast_rows": (_I, [_P, _I, _P, _I, _I, _I, _I, _P]),
    "hhrl_copy2d": (_I, [_P, _I, _P, _I, _I, _I, _P]),
    "hhrl_copy2d_group": (_I, [_P, _I, _I, _P]),
    "hhrl_gather_rows": (_I, [_P, _L, _I, _P, _P, _I, _I, _I, _P]),
    "hhrl_gather_copy_group": (_I, [_P, _I, _I, _P, _I, _P]),
    "hhrl_fill": (_I, [_P, _L, _F, _P]),
    "hhrl_randn_fill": (_I, [_P, _L, C.c_ulonglong, C.c_ulonglong, _P]),
    "hhrl_vae_reparam": (_I, [_P, _P, _P, _I, _I, _P]),
    "hhrl_vae_elbo": (_I, [_P, _P, _P, _F, _P, _P, _I, _I, _I, _P]),
    "hhrl_vae_latent_grad": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "hhrl_vae_workspace_floats": (_L, [_I, _I, _I, _L]),
    "hhrl_vae_train_step": (_I, [_P, _P, _P, _L, _P, _P, _I, _I, _I, _F, _P, _P, _P]),
    "hhrl_vae_train_step_noalloc": (_I, [_P, _P, _P, _L, _P, _P, _I, _I, _I, _F, _P, _P]),
    "hhrl_vae_learn_step": (_I, [_P, _P, _P, _P, _P, _L, _P, _P, _I, _I, _I, _F, _D, _D, _D, _D, _I, _P, _P, _P]),
    "hhrl_vae_learn_steps": (_I, [_P, _P, _P, _P, _P, _L, _I, _P, _P, _P, _P, _I, _D, _D, _D, _D, _I, _P, _P, _P]),
    "hhrl_vae_encode": (_I, [_P, _P, _L, _P, _P, _I, _I, _I, _P, _I, _P]),
    "hhrl_vae_decode": (_I, [_P, _P, _L, _P, _I, _P, _P, _I, _I, _I, _I, _P, _I, _P]),
    "hhrl_td3_smooth_action": (_I, [_P, _I, _P, _P, _I, _I, _I, _P]),
    "hhrl_td3_target_q": (_I, [_P, _P, _P, _I, _F, _P, _I, _P]),
    "hhrl_twin_mse": (_I, [_P, _P, _P, _F, _P, _P, _P, _I, _P]),
    "hhrl_td3_target_mse": (_I, [_P, _P, _P, _I, _F, _P, _P, _F, _P, _P, _P, _P, _I, _P]),
    "hhrl_td3_target_mse_zeroed": (_I, [_P, _P, _P, _I, _F, _P, _P, _F, _P, _P, _P, _P, _I, _P]),
    "hhrl_lstm_workspace_floats": (_L, [_I, _I, _I, _I]),
    "hhrl_lstm_forward": (_I, [_P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _P]),
    "hhrl_lstm_backward": (_I, [_P, _P, _I, _P, _P, _I, _I, _I, _I, _P]),
    "hhrl_lstm_forward_group": (_I, [_P, _I, _I, _I, _I, _P]),
    "hhrl_lstm_backward_group": (_I, [_P, _I, _I, _I, _I, _P]),
    "hhrl_adam_step": (_I, [_P, _P, _P, _P, _L, _D, _D, _D, _D, _I, _D, _P]),
    "hhrl_adam_step_copy": (_I, [_P, _P, _P, _P, _L, _D, _D, _D, _D, _I, _D, _P, _P, _P]),
    "hhrl_adam_stage": (_I, [_P, _D, _D, _D, _D, _I, _D, _P]),
    "hhrl_adam_step_dev": (_I, [_P, _P, _P, _P, _L, _P, _P]),
    "hhrl_adam_polyak_step_dev": (_I, [_P, _P, _P, _P, _L, _P, _P, _D, _P]),
    "hhrl_polyak_update": (_I, [_P, _P, _L, _D, _P]),
    "hhrl_prof_enable": (_I, [_I]),
    "hhrl_prof_reset": (_I, []),
    "hhrl_prof_report": (_L, [C.c_char_p, _L]),
}

_lib = None


def load() -> C.CDLL:
    """Load the shared library (once) and declare every prototype."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("HHRL_HIP_LIB", LIB_PATH)      # diagnostic builds of the same sources
    if not os.path.exists(path):
        raise RuntimeError(
            "HIP extension not built: %s is missing. Run `python hhrl-vae_amd/build.py` "
            "(needs hipcc; cross-compiles for gfx950 without a GPU)." % path)
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.hhrl_version() != ABI_VERSION:
        raise RuntimeError("%s was built for ABI version %d, these bindings are written for %d (the descriptor structs differ): "
                           "rebuild it with `python hhrl-vae_amd/build.py`" % (path, lib.hhrl_version(), ABI_VERSION))
    _lib = lib
    return lib


class HipError(RuntimeError):
    pass


_hip_rt = None
ZERO_COPY_MAX_FLOATS = 1 << 16


def zero_copy_ok(pinned, any_size: bool = False) -> bool:
    """True when kernels may read and write this pinned host tensor in place, at its host address: the HIP runtime
    reports a device address for it and that address is the host one (unified addressing; torch's pinned memory is
    mapped).  The per-call inference paths then hand their pinned blocks straight to the kernels instead of staging
    them through device buffers (one launch instead of copy + launch + copy).  HHRL_ZERO_COPY=0 turns it off; blocks
    beyond ZERO_COPY_MAX_FLOATS are staged (a DMA copy moves large blocks faster than a kernel's own reads) unless
    the caller reads only a small part of the block (any_size)."""
    global _hip_rt
    if os.environ.get("HHRL_ZERO_COPY") == "0" or not pinned.is_pinned() or (pinned.numel() > ZERO_COPY_MAX_FLOATS and not any_size):
        return False
    try:
        if _hip_rt is None:
            _hip_rt = C.CDLL("libamdhip64.so")
        dptr = C.c_void_p()
        rc = _hip_rt.hipHostGetDevicePointer(C.byref(dptr), C.c_void_p(pinned.data_ptr()), 0)
    except OSError:
        return False
    return rc == 0 and dptr.value == pinned.data_ptr()


def check(rc: int) -> None:
    if rc != 0:
        msg = load().hhrl_last_error()
        raise HipError("libhhrl_hip error %d: %s" % (rc, msg.decode() if msg else "?"))


_raw_stream = None
_streams = {}


def stream_ptr() -> int:
    """The raw hipStream_t of torch's current stream: kernels are enqueued
    there so they order with torch's own copies and allocations.  Read through torch's C accessor:
    `torch.cuda.current_stream()` builds a Stream object and resolves the device each time, ten
    microseconds that every entry-point call and every update would pay."""
    global _raw_stream
    import torch
    if _raw_stream is None:
        _raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", False)
    if _raw_stream:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def current_stream():
    """torch's current stream as a `torch.cuda.Stream`, cached per (device, raw handle)."""
    import torch
    key = (torch.cuda.current_device(), stream_ptr())
    st = _streams.get(key)
    if st is None:
        if len(_streams) > 64:
            _streams.clear()
        st = _streams[key] = torch.cuda.current_stream()
    return st


def prof_report() -> dict:
    """{kernel name: {"launches", "ms", "flops", "bytes"}} since the last reset
    (synchronises on the recorded events)."""
    lib = load()
    n = lib.hhrl_prof_report(None, 0)
    buf = C.create_string_buffer(int(n) + 16)
    lib.hhrl_prof_report(buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        name, cnt, ms, fl, by = line.rsplit(" ", 4)
        out[name] = {"launches": int(cnt), "ms": float(ms), "flops": float(fl), "bytes": float(by)}
    return out
