#!/usr/bin/env python3
"""Per-tensor distance of the GPU gradients (fused and per-layer paths) and of
the fp32 CPU oracle from an fp64 evaluation of the same step."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "golden")]
import numpy as np
import torch

import gen
import hhrl_oracle as O
from hhrl_vae_amd.algorithm import Encoder


def main():
    B, N, g = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (4096, 30, 4)
    shapes = O.vae_param_shapes(g)
    init = gen.gen_state(shapes, 1234 + g)
    st = {k: torch.from_numpy(v) for k, v in init.items()}
    x = gen.gen_lanes(B, N, 78)
    eps = torch.randn(B, g, generator=torch.Generator().manual_seed(3408))
    o32 = O.VAE(dict(st), g)
    o64 = O.VAE(dict(st), g, dtype=torch.float64)
    _, g32 = o32.loss_and_grads(torch.from_numpy(x), eps)
    _, g64 = o64.loss_and_grads(torch.from_numpy(x), eps)
    res = {}
    for fused in (True, False):
        e = Encoder(2, g, device="cuda:0", seed=1, fused=fused)
        e.load_state_dict(st)
        e.loss_and_grads(torch.from_numpy(x).cuda(), eps=eps)
        res[fused] = {k: v.detach().cpu().numpy() for k, v in e.grads().items()}
    print("%-28s %10s %12s %12s %12s" % ("tensor", "max|g|", "oracle32", "fused", "per-layer"))
    for k in g64:
        ref = g64[k].numpy()
        sc = np.abs(ref).max()
        d = lambda a: np.abs(np.asarray(a, dtype=np.float64) - ref).max()
        print("%-28s %10.3e %12.3e %12.3e %12.3e" % (k, sc, d(g32[k].numpy()), d(res[True][k]), d(res[False][k])))


if __name__ == "__main__":
    main()
