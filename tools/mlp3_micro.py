#!/usr/bin/env python3
"""Micro-benchmark of the fused three-layer kernels through the C ABI: n networks (K0=18, N3=1, the worker
critic's heads) on M rows, forward and backward launched back to back.  Prints the time per launch from
HIP events around the loop; under `rocprofv3 --kernel-trace --stats` the trace gives the kernels' own durations.

    python tools/mlp3_micro.py M n [reps]
"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hhrl_vae_amd import _lib as L  # noqa: E402

M, n = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
lib = L.load()
dev = torch.device("cuda:0")
K0, N3 = 18, 1
descs = (L.Mlp3Desc * n)()
keep = []
g = torch.Generator().manual_seed(1)
for i in range(n):
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    x = r(M, K0)
    ws = [r(128, K0) / K0 ** 0.5, r(128), r(128, 128) / 128 ** 0.5, r(128), r(N3, 128) / 128 ** 0.5, r(N3)]
    gd = [torch.zeros_like(w) for w in ws]
    h1, h2, y, dy = torch.empty(M, 128, device=dev), torch.empty(M, 128, device=dev), torch.empty(M, N3, device=dev), r(M, N3)
    d = descs[i]
    d.x, d.ldx, d.K0 = x.data_ptr(), K0, K0
    d.w1, d.b1, d.w2, d.b2, d.w3, d.b3 = [w.data_ptr() for w in ws]
    d.N3, d.act3, d.h1, d.h2, d.y, d.ldy = N3, 0, h1.data_ptr(), h2.data_ptr(), y.data_ptr(), N3
    d.dy, d.lddy, d.dy_const = dy.data_ptr(), N3, 0.0
    d.gw1, d.gb1, d.gw2, d.gb2, d.gw3, d.gb3 = [t.data_ptr() for t in gd]
    n_slab = lib.hhrl_mlp3_slab_floats(K0, N3, M)
    slab = torch.empty(max(n_slab, 1), device=dev)
    if n_slab and os.environ.get("MLP3_NO_SLAB") != "1":
        d.gslab, d.gslab_floats = slab.data_ptr(), n_slab
    keep.append((x, ws, gd, h1, h2, y, dy, slab))
st = L.stream_ptr()


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps


f = timed(lambda: L.check(lib.hhrl_mlp3_forward_group(descs, n, M, st)))
b = timed(lambda: L.check(lib.hhrl_mlp3_backward_group(descs, n, M, st)))
print("M=%d n=%d WGS=%s: forward %.2f us, backward %.2f us per launch (back to back)" % (M, n, os.environ.get("HHRL_MLP3_WGS", "256"), f, b))
