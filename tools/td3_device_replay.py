#!/usr/bin/env python3
"""Update time with the replay buffer gathered on the host (default) against device_replay=True (rows
mirrored in device memory, minibatch gathered there), one stored transition per update as in the control
loop.  Run on a GPU box: python tools/td3_device_replay.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch, gen
from hhrl_vae_amd.algorithm import HATD3Triple, WorkerTD3
for name, mk, n in (("worker", lambda dr: WorkerTD3(gen.cav_cfg(cap=4096), device_replay=dr), 4096),
                    ("triple", lambda dr: HATD3Triple(gen.light_cfg(cap=2048), device_replay=dr), 2048)):
    for dr in (False, True):
        net = mk(dr)
        net.memory[:n] = np.random.RandomState(1).uniform(0, 1, size=(n, net.fam.row_width)); net.pointer = n
        for it in range(2000):                      # until every launch chain is replayed from its graphs
            net.learn() if name == "worker" else net.learn(None)
            f = net.launch_forms()
            if it >= 30 and len(f) == (2 if name == "worker" else 7) and "tuning" not in f.values(): break
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < 0.5:      # ... and the device has been busy for half a second
            net.learn() if name == "worker" else net.learn(None)
        torch.cuda.synchronize()
        row = np.random.RandomState(2).uniform(0, 1, size=net.fam.row_width)
        t0 = time.perf_counter()
        for _ in range(1000):
            net._store(row)                         # one new transition per update, as in the control loop
            net.learn() if name == "worker" else net.learn(None)
        torch.cuda.synchronize()
        print(name, "device_replay", dr, "ms/update %.4f" % (1e3 * (time.perf_counter() - t0) / 1000))
