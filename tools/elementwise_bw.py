#!/usr/bin/env python3
"""Achieved HBM bandwidth of the optimizer kernels against the chip's peak (8 TB/s): Adam with the
scalars as kernel arguments (hhrl_adam_step), Adam with the scalars in device memory (hhrl_adam_step_dev,
the form inside captured update graphs) and the Polyak target update, at the parameter counts of the
reference's networks and at sizes that leave the caches.  Algorithmic bytes: Adam 28 B / parameter
(reads p, g, m, v, writes p, m, v), Polyak 12 B / parameter.  Timed with events on the launch stream,
`reps` launches back to back.

    python tools/elementwise_bw.py [--json out.json]        (on a GPU box)
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

PEAK_GBS = 8000.0
SIZES = [("VAE g=4", 30905), ("WorkerCritic", 38146), ("ManagerActor", 364720), ("TripleCritic", 1216898),
         ("16 Mi", 1 << 24), ("100 M", 100000000), ("128 Mi", 1 << 27)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    from hhrl_vae_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    st = _lib.stream_ptr()
    rows = []
    for name, n in SIZES:
        p, g, m, v, t = (torch.rand(n, device=dev, dtype=torch.float32) for _ in range(5))
        hyper = torch.zeros(8, device=dev, dtype=torch.float32)
        _lib.check(lib.hhrl_adam_stage(hyper.data_ptr(), 1e-3, 0.9, 0.999, 1e-8, 1, 1.0, st))
        calls = {
            "adam_step": (28.0, lambda: lib.hhrl_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n,
                                                           1e-3, 0.9, 0.999, 1e-8, 1, 1.0, st)),
            "adam_step_dev": (28.0, lambda: lib.hhrl_adam_step_dev(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n,
                                                                   hyper.data_ptr(), st)),
            "polyak_update": (12.0, lambda: lib.hhrl_polyak_update(t.data_ptr(), p.data_ptr(), n, 0.005, st)),
        }
        reps = 200 if n < (1 << 22) else 20
        for kname, (bpp, fn) in calls.items():
            for _ in range(3):
                _lib.check(fn())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / reps
            gbs = bpp * n / (us * 1e-6) / 1e9
            rows.append({"kernel": kname, "network": name, "params": n, "us_per_launch": us, "GB_per_s": gbs,
                         "frac_of_hbm_peak": gbs / PEAK_GBS})
            print("%-14s %-13s n=%10d  %9.2f us  %8.1f GB/s  %.3f of 8 TB/s" % (kname, name, n, us, gbs, gbs / PEAK_GBS))
    if args.json:
        with open(args.json, "w") as f:
            json.dump({"peak_GB_per_s": PEAK_GBS, "rows": rows}, f, indent=1)
            f.write("\n")


if __name__ == "__main__":
    main()
