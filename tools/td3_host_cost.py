#!/usr/bin/env python3
"""Where a batch-64 update spends host time: replay gather + upload vs the rest."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden")]
import numpy as np, torch, gen
from hhrl_vae_amd.algorithm import HATD3Triple, WorkerTD3

for name, net, n in (("worker", WorkerTD3(gen.cav_cfg(cap=120000)), 120000), ("triple", HATD3Triple(gen.light_cfg(cap=160000)), 160000)):
    net.memory[:n] = np.random.RandomState(1).uniform(0, 1, size=(n, net.fam.row_width)); net.pointer = n
    for _ in range(40): net.learn()
    torch.cuda.synchronize()
    K = 200
    t0 = time.perf_counter()
    for _ in range(K): net.learn()
    torch.cuda.synchronize()
    full = (time.perf_counter() - t0) / K
    t0 = time.perf_counter()
    for _ in range(K):
        idx = np.random.choice(n, size=net.batch_size)
        rows = net._sample_rows(idx)
    gather = (time.perf_counter() - t0) / K
    st = net._statics[net.batch_size]
    ring = next(iter(st.rings.values()))
    t0 = time.perf_counter()
    for _ in range(K): st.d_in.copy_(ring.h_in[0][:st.total], non_blocking=True)
    torch.cuda.synchronize()
    h2d = (time.perf_counter() - t0) / K
    rows_d = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.float32)).cuda()
    t0 = time.perf_counter()
    for _ in range(K): net.learn_rows(rows_d)
    torch.cuda.synchronize()
    dev = (time.perf_counter() - t0) / K
    print("%s: learn() %.1f us; host gather %.1f us; pinned H2D of rows+noise+scalars %.1f us; learn_rows(device rows) %.1f us" % (name, 1e6 * full, 1e6 * gather, 1e6 * h2d, 1e6 * dev))
