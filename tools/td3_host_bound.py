#!/usr/bin/env python3
"""Is an update chain bound by the host or by the GPU?  K updates are enqueued; the time the host spends waiting
for a free staging block (every launch chain has two) is set against the time of the whole stretch.
No waiting: the host is the bound.  Waiting = total - host work: the GPU is.

    python tools/td3_host_bound.py <worker|triple> <batch> [updates]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden")]
import numpy as np
import torch

import gen
from hhrl_vae_amd.algorithm import HATD3Triple, WorkerTD3


def main():
    which, batch = sys.argv[1], int(sys.argv[2])
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
    n_rows = 8192 if which == "worker" else 2048
    net = WorkerTD3(gen.cav_cfg(cap=n_rows, batch=batch)) if which == "worker" else HATD3Triple(gen.light_cfg(cap=n_rows, batch=batch))
    net.memory[:n_rows] = np.random.RandomState(3407).uniform(0, 1, size=(n_rows, net.fam.row_width))
    net.pointer = n_rows
    t_w = time.perf_counter()
    it = 0
    while time.perf_counter() - t_w < 1.5 or it < 100:
        net.learn()
        it += 1
    torch.cuda.synchronize()
    from hhrl_vae_amd.td3 import TD3Learner
    waited = [0.0]
    plain = TD3Learner._Ring.next_slot

    def timed_slot(ring):
        t = time.perf_counter()
        slot = plain(ring)
        waited[0] += time.perf_counter() - t
        return slot

    TD3Learner._Ring.next_slot = timed_slot
    learn = net.learn
    if os.environ.get("TD3_NO_HOST_RNG") == "1":        # experiment: no draws on the host (fixed row numbers, stale noise)
        class _NoDraw:
            def standard_normal(self, *a, **k):
                return None
        net._np_rng = _NoDraw()
        idx = np.random.RandomState(1).choice(n_rows, size=batch)
        learn = lambda: net.learn(indices=idx)
    for trial in range(3):
        waited[0] = 0.0
        t0 = time.perf_counter()
        for _ in range(K):
            learn()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("%s b%d: %.1f us per update, of which the host waited %.1f us for the device and worked %.1f us" %
              (which, batch, 1e6 * (t2 - t0) / K, 1e6 * waited[0] / K, 1e6 * (t2 - t0 - waited[0]) / K))


if __name__ == "__main__":
    main()
