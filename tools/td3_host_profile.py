#!/usr/bin/env python3
"""cProfile of the host side of K updates (the device work is enqueued, not waited for, inside the profile).

    python tools/td3_host_profile.py <worker|triple> <batch> [updates]"""
import cProfile
import os
import pstats
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
    while time.perf_counter() - t_w < 1.0:
        net.learn()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(K):
        net.learn()
    pr.disable()
    torch.cuda.synchronize()
    ps = pstats.Stats(pr)
    ps.sort_stats("tottime").print_stats(22)


if __name__ == "__main__":
    main()
