#!/usr/bin/env python3
"""A fixed number of TD3 updates of one configuration, for a profiler to trace:

    rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 tools/td3_trace.py <worker|triple> <batch> [updates]

Replay memory is synthetic (uniform rows); the updates run in the form the class settles on (HIP-graph replay at
these sizes), after a warm-up that lets every launch chain reach that form.  Prints one JSON line with the
wall time per update of the traced stretch."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import torch

import gen
from hhrl_vae_amd.algorithm import HATD3Triple, WorkerTD3


def main():
    which, batch = sys.argv[1], int(sys.argv[2])
    updates = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    n_rows = 8192 if which == "worker" else 2048
    net = WorkerTD3(gen.cav_cfg(cap=n_rows, batch=batch)) if which == "worker" else HATD3Triple(gen.light_cfg(cap=n_rows, batch=batch))
    net.memory[:n_rows] = np.random.RandomState(3407).uniform(0, 1, size=(n_rows, net.fam.row_width))
    net.pointer = n_rows
    n_chains = 2 if which == "worker" else 7
    for it in range(800):
        net.learn()
        f = net.launch_forms()
        if it >= 30 and len(f) == n_chains and "tuning" not in f.values():
            break
    # ... and until the device has been busy with this update for half a second, as bench.py does: a chain bound by
    # the GPU is then timed at the clocks the chip settles on
    torch.cuda.synchronize()
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < (0.0 if os.environ.get("TD3_TRACE_BRIEF") == "1" else 0.5):    # (counter passes: brief)
        net.learn()
        it += 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(updates):
        net.learn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"config": "%s_b%d" % (which, batch), "updates": updates, "warmup_updates": it + 1,
                      "ms_per_update": 1e3 * dt / updates, "launch_forms": sorted(set(net.launch_forms().values()))}))


if __name__ == "__main__":
    main()
