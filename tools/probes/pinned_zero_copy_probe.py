#!/usr/bin/env python3
"""Can a kernel of the library read and write torch's pinned host memory directly (no staging copy)?
Asks the HIP runtime for the device address of a pinned tensor (hipHostGetDevicePointer; nothing is launched unless
that succeeds), fills the tensor from a kernel, reads it on the host, and times a small kernel + synchronize with
pinned and with device buffers."""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hhrl_vae_amd import _lib as L  # noqa: E402

lib = L.load()
hip = C.CDLL("libamdhip64.so")
torch.cuda.init()
h = torch.zeros(1024, dtype=torch.float32).pin_memory()
dptr = C.c_void_p()
rc = hip.hipHostGetDevicePointer(C.byref(dptr), C.c_void_p(h.data_ptr()), 0)
print("hipHostGetDevicePointer rc=%d host=%#x device=%s" % (rc, h.data_ptr(), hex(dptr.value) if dptr.value else None))
if rc != 0 or not dptr.value:
    sys.exit(0)
st = L.stream_ptr()
L.check(lib.hhrl_fill(dptr.value, 1024, 3.5, st))
torch.cuda.synchronize()
print("host sees", float(h[0]), float(h[1023]), "after a device fill of the pinned tensor")
d = torch.zeros(1024, device="cuda")
hh = torch.zeros(1024).pin_memory()


def timed(fn, n=2000):
    for _ in range(200):
        fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return 1e6 * (time.perf_counter() - t0) / n


def staged():
    L.check(lib.hhrl_fill(d.data_ptr(), 64, 1.0, st))
    hh.copy_(d, non_blocking=True)
    torch.cuda.current_stream().synchronize()


def direct():
    L.check(lib.hhrl_fill(dptr.value, 64, 1.0, st))
    torch.cuda.current_stream().synchronize()


print("kernel + copy to pinned + synchronize: %.1f us; kernel writing pinned memory + synchronize: %.1f us" % (timed(staged), timed(direct)))
