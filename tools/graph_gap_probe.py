#!/usr/bin/env python3
"""Dead time between dependent kernels: a chain of K fused-MLP forward launches (each reads what the one before
wrote, 13 us of device work, far more than a launch costs the host) enqueued on the stream against the same chain
captured once into a HIP graph and replayed.  Device time per launch from events around a long run.

    python tools/graph_gap_probe.py [M] [n] [chain]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hhrl_vae_amd import _lib as L  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
chain = int(sys.argv[3]) if len(sys.argv) > 3 else 8
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
    h1, h2 = torch.empty(M, 128, device=dev), torch.empty(M, 128, device=dev)
    d = descs[i]
    d.x, d.ldx, d.K0 = x.data_ptr(), K0, K0
    d.w1, d.b1, d.w2, d.b2, d.w3, d.b3 = [w.data_ptr() for w in ws]
    # the output lands in the first input column: the next launch of the chain depends on this one
    d.N3, d.act3, d.h1, d.h2, d.y, d.ldy = N3, 1, h1.data_ptr(), h2.data_ptr(), x.data_ptr(), K0
    keep.append((x, ws, h1, h2))


def run_chain():
    st = L.stream_ptr()
    for _ in range(chain):
        L.check(lib.hhrl_mlp3_forward_group(descs, n, M, st))


def timed(fn, reps):
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.5:
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / (reps * chain)


s = torch.cuda.Stream()
with torch.cuda.stream(s):
    run_chain()
    torch.cuda.synchronize()
    eager = timed(run_chain, 200)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        run_chain()
    replay = timed(graph.replay, 200)
print("M=%d n=%d chain of %d: %.2f us per launch enqueued on the stream, %.2f us per launch replayed from a graph" % (M, n, chain, eager, replay))
