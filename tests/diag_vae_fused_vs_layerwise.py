"""Diagnostic (not collected by pytest): how the fused VAE step and the one-launch-per-layer path differ, per gradient
tensor, at the sizes where test_vae_fused_matches_layerwise needs more than the 1e-5 bar: the largest difference
relative to the tensor's largest entry, how many elements and how many rows (first axis) lie beyond 1e-5 / 2e-5 / 5e-5.

    python tests/diag_vae_fused_vs_layerwise.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import gen  # noqa: E402
import hhrl_oracle as O  # noqa: E402
from hhrl_vae_amd.algorithm import Encoder  # noqa: E402

for B, N, g in [(4096, 7, 4), (4096, 30, 4), (2048, 59, 8), (9000, 2, 4), (20000, 7, 4), (1000, 3, 16)]:
    init = gen.gen_state(O.vae_param_shapes(g), 500 + B + N)
    ef, el = Encoder(2, g, fused=True), Encoder(2, g, fused=False)
    ef.load_state_dict(init)
    el.load_state_dict(init)
    x = torch.from_numpy(gen.gen_lanes(B, N, 9)).cuda()
    eps = torch.randn(B, g, generator=torch.Generator().manual_seed(B * N)).cuda()
    ef.loss_and_grads(x, eps)
    el.loss_and_grads(x, eps)
    gf, gl = ef.grads(), el.grads()
    print("B=%d N=%d g=%d" % (B, N, g))
    for k in gf:
        a, b = gf[k].numpy().astype(np.float64), gl[k].numpy().astype(np.float64)
        m = np.abs(b).max()
        d = np.abs(a - b) / max(m, 1e-30)
        d2 = d.reshape(d.shape[0], -1)
        rows = lambda t: int((d2.max(axis=1) > t).sum())
        print("  %-24s %-12s max %.2e  elems>1e-5 %5d  >2e-5 %5d  >5e-5 %5d   rows>1e-5 %4d >2e-5 %4d >5e-5 %4d of %d"
              % (k, tuple(a.shape), d.max(), (d > 1e-5).sum(), (d > 2e-5).sum(), (d > 5e-5).sum(), rows(1e-5), rows(2e-5), rows(5e-5),
                 d2.shape[0]))
