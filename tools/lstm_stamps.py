#!/usr/bin/env python3
"""Phase breakdown inside the one-launch LSTM recurrence (diagnostic build).

Build first (CPU box): python hhrl-vae_amd/build.py --stamps
Run (GPU box):         python tools/lstm_stamps.py [B T n]

Workgroup (0,0,0) / thread 0 records the shader clock at marked points; prints distances in clocks."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["HHRL_HIP_LIB"] = os.path.join(ROOT, "hhrl-vae_amd", "lib", "libhhrl_hip_stamps.so")
import numpy as np
import torch

from hhrl_vae_amd import _lib as L

FWD = [(0, "start"), (8, "W_ih requested"), (9, "first slab of W_hh in registers"), (10, "all slabs"), (11, "observations staged"), (1, "requests complete, barrier"), (2, "step 1: products"), (3, "step 1: exchange + cell"),
       (4, "step 1: stores issued"), (5, "step 1: barrier"), (6, "step 2: products"), (7, "end of last step")]
BWD = [(0, "start"), (1, "weights in registers, last cell"), (6, "first step (all four phases)"), (2, "second step: products"),
       (3, "second step: barrier"), (4, "second step: sum + cell"), (5, "second step: barrier"), (7, "end of last step")]


def main():
    B, T, n = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (64, 10, 9)
    H, dev = 128, "cuda:0"
    lib = L.load()
    obs_dims = ([13, 13, 25] * 6)[:n]
    descs = (L.LstmDesc * n)()
    keep = []
    g = torch.Generator().manual_seed(1)
    for i, o in enumerate(obs_dims):
        shapes = [(4 * H, o), (4 * H, H), (4 * H,), (4 * H,), (4 * H, o), (4 * H, H), (4 * H,), (4 * H,), (H, 2 * H), (H,)]
        pd = [(0.1 * torch.randn(*s, generator=g)).to(dev) for s in shapes]
        parr = (C.c_void_p * 10)(*[p.data_ptr() for p in pd])
        ws = torch.empty(lib.hhrl_lstm_workspace_floats(B, T, o, H), device=dev)
        od = torch.zeros(B, H, device=dev)
        obs = torch.randn(B, T * o, generator=g).to(dev)
        d = descs[i]
        d.obs, d.ldobs, d.o = obs.data_ptr(), T * o, o
        d.params = C.cast(parr, C.c_void_p)
        d.out, d.ldo, d.ws = od.data_ptr(), H, ws.data_ptr()
        d.no_backward = 1 if os.environ.get("LSTM_NO_BACKWARD") == "1" else 0      # (forward stamps without the saved steps)
        keep.append((pd, parr, ws, od, obs))
    st = L.stream_ptr()
    for _ in range(5):
        L.check(lib.hhrl_lstm_forward_group(descs, n, B, T, H, st))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        L.check(lib.hhrl_lstm_forward_group(descs, n, B, T, H, st))
    e1.record()
    torch.cuda.synchronize()
    print("forward group (sequence kernel + fc product): %.1f us per call" % (1e3 * e0.elapsed_time(e1) / 20))
    buf = (C.c_longlong * 128)()
    lib.hhrl_debug_lstm_stamps.restype = C.c_int
    assert lib.hhrl_debug_lstm_stamps(buf) == 0
    s = np.array(buf[:]).reshape(2, 64)
    prev = s[0, 0]
    for idx, label in FWD[1:]:
        print("  %-36s %8d" % (label, s[0, idx] - prev))
        prev = s[0, idx]
    print("  whole kernel (workgroup 0): %d clocks" % (s[0, 7] - s[0, 0]))
    # backward
    for i in range(n):
        d = descs[i]
        o = obs_dims[i]
        shapes = [(4 * H, o), (4 * H, H), (4 * H,), (4 * H,), (4 * H, o), (4 * H, H), (4 * H,), (4 * H,), (H, 2 * H), (H,)]
        gd = [torch.zeros(*sh, device=dev) for sh in shapes]
        garr = (C.c_void_p * 10)(*[p.data_ptr() for p in gd])
        dout = torch.randn(B, H, generator=g).to(dev)
        d.grads = C.cast(garr, C.c_void_p)
        d.d_out, d.lddo = dout.data_ptr(), H
        keep.append((gd, garr, dout))
    for _ in range(5):
        L.check(lib.hhrl_lstm_backward_group(descs, n, B, T, H, st))
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        L.check(lib.hhrl_lstm_backward_group(descs, n, B, T, H, st))
    e1.record()
    torch.cuda.synchronize()
    print("backward group (fc products, sequence kernel, weight-gradient products, bias sums): %.1f us per call" % (1e3 * e0.elapsed_time(e1) / 20))
    assert lib.hhrl_debug_lstm_stamps(buf) == 0
    s = np.array(buf[:]).reshape(2, 64)
    prev = s[1, 0]
    for idx, label in BWD[1:]:
        print("  %-36s %8d" % (label, s[1, idx] - prev))
        prev = s[1, idx]
    print("  whole kernel (workgroup 0): %d clocks" % (s[1, 7] - s[1, 0]))


if __name__ == "__main__":
    main()
