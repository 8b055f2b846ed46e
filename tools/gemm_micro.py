#!/usr/bin/env python3
"""Micro-benchmark of the layer products through the C ABI: n independent layers (M rows, K inputs, N outputs,
ReLU) per launch, forward and backward (input + weight + bias gradient, activation gradient inside), launched
back to back.  Prints the time per launch from HIP events around the loop and the rate by the algorithm's
count.  HHRL_GEMM_BLOCK_MIN=0 keeps the 32 x 32 tile kernels for an A/B.

    python tools/gemm_micro.py "M K N n" ["M K N n" ...]   (ld: optional 5th value, leading dimension of x)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hhrl_vae_amd import _lib as L  # noqa: E402

lib = L.load()
dev = torch.device("cuda:0")
st = L.stream_ptr()
reps = int(os.environ.get("REPS", "50"))


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


for spec in sys.argv[1:]:
    v = [int(t) for t in spec.split()]
    M, K, N, n = v[:4]
    ldx = v[4] if len(v) > 4 else K
    g = torch.Generator().manual_seed(1)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    fd, bd, keep = (L.LinearFwdDesc * n)(), (L.LinearBwdDesc * n)(), []
    for i in range(n):
        x, w, b, y, dy = r(M, ldx), r(N, K) / K ** 0.5, r(N), torch.empty(M, N, device=dev), r(M, N)
        dx, dw, db = torch.empty(M, ldx, device=dev), torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
        f = fd[i]
        f.x, f.ldx, f.w, f.b, f.y, f.ldy, f.K, f.N, f.act, f.accumulate = x.data_ptr(), ldx, w.data_ptr(), b.data_ptr(), y.data_ptr(), N, K, N, 1, 0
        q = bd[i]
        q.x, q.ldx, q.w, q.y, q.ldy, q.dy, q.lddy = x.data_ptr(), ldx, w.data_ptr(), y.data_ptr(), N, dy.data_ptr(), N
        q.dx, q.lddx, q.dx_accumulate, q.dw, q.db, q.K, q.N, q.act = dx.data_ptr(), ldx, 0, dw.data_ptr(), db.data_ptr(), K, N, 1
        keep.append((x, w, b, y, dy, dx, dw, db))
    tf = timed(lambda: L.check(lib.hhrl_linear_forward_group(fd, n, M, st)))
    tb = timed(lambda: L.check(lib.hhrl_linear_backward_group(bd, n, M, st)))
    fl = 2.0 * M * K * N * n
    print("M=%d K=%d N=%d n=%d ldx=%d block_min=%s: forward %.2f us (%.1f TFLOP/s), backward %.2f us (%.1f TFLOP/s)" % (
        M, K, N, n, ldx, os.environ.get("HHRL_GEMM_BLOCK_MIN", "default"), tf, fl / tf * 1e-6, tb, 2 * fl / tb * 1e-6), flush=True)
