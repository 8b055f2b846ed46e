#!/usr/bin/env python3
"""Micro-benchmark of the grouped BiLSTM entry points through the C ABI: n modules (T = 10, H = 128, observation
widths cycling through 13, 13, 25) on B rows; forward and backward launched back to back.  Prints the time per call
from HIP events around the loop (median of five loops) and the forward rate by the algorithm's count.
(Environment switches of the library, e.g. HHRL_LSTM_SEQ=0, apply: run it once per setting for an A/B.)

    python tools/lstm_micro.py "n B" ["n B" ...]
"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hhrl_vae_amd import _lib as L  # noqa: E402

lib = L.load()
dev = torch.device("cuda:0")
st = L.stream_ptr()
T, H = 10, 128
reps = int(os.environ.get("REPS", "30"))


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        out.append(1e3 * a.elapsed_time(b) / reps)
    return sorted(out)[2], out


for spec in sys.argv[1:]:
    n, B = [int(t) for t in spec.split()]
    g = torch.Generator().manual_seed(1)
    r = lambda *s: (torch.randn(*s, generator=g) * 0.3).to(dev)
    descs, keep, macs = (L.LstmDesc * n)(), [], 0
    for i in range(n):
        o = (13, 13, 25)[i % 3]
        ps = [r(4 * H, o), r(4 * H, H), r(4 * H), r(4 * H), r(4 * H, o), r(4 * H, H), r(4 * H), r(4 * H), r(H, 2 * H), r(H)]
        gd = [torch.zeros_like(p) for p in ps]
        parr = (C.c_void_p * 10)(*[p.data_ptr() for p in ps])
        garr = (C.c_void_p * 10)(*[p.data_ptr() for p in gd])
        ws = torch.empty(lib.hhrl_lstm_workspace_floats(B, T, o, H), device=dev)
        obs, od, dout = r(B, T * o), torch.zeros(B, H, device=dev), r(B, H)
        d = descs[i]
        d.obs, d.ldobs, d.o = obs.data_ptr(), T * o, o
        d.params, d.grads = C.cast(parr, C.c_void_p), C.cast(garr, C.c_void_p)
        d.out, d.ldo, d.d_out, d.lddo, d.ws = od.data_ptr(), H, dout.data_ptr(), H, ws.data_ptr()
        keep.append((ps, gd, parr, garr, ws, obs, od, dout))
        macs += B * (T * 4 * H * (o + H) + 4 * H * (o + H) + 2 * H * H)
    tf, allf = timed(lambda: L.check(lib.hhrl_lstm_forward_group(descs, n, B, T, H, st)))
    tb, _ = timed(lambda: L.check(lib.hhrl_lstm_backward_group(descs, n, B, T, H, st)))
    print("n=%d B=%d: forward %.1f us (%.1f TFLOP/s; loops %s), backward %.1f us" % (
        n, B, tf, 2.0 * macs / tf * 1e-6, " ".join("%.0f" % v for v in allf), tb), flush=True)
