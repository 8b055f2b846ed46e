#!/usr/bin/env python3
"""Phase breakdown inside the fused VAE kernels (diagnostic build).

Build first (CPU box): python hhrl-vae_amd/build.py --stamps
Run (GPU box):         python tools/vae_stamps.py [B N g]

Workgroup 0 / thread 0 records the shader clock at marked points; the script
prints the distance between consecutive stamps in clocks and as a share of
the kernel's span."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["HHRL_HIP_LIB"] = os.path.join(ROOT, "hhrl-vae_amd", "lib", "libhhrl_hip_stamps.so")

import numpy as np
import torch

from hhrl_vae_amd import _lib
from hhrl_vae_amd.algorithm import Encoder

CAR = ["start", "weights in LDS", "tile inputs", "func.0 fwd (element-wise)", "func.2 fwd", "func.4+err", "func.4 bwd",
       "func.2 bwd products", "mask, column sums, df1 image", "per-lane sums of df1", "pre-combine barrier",
       "per-wave buffers in LDS", "slab written"]
CHAIN = ["start", "weights in LDS", "cars staged", "phi.0", "phi.2", "rho.0", "rho.2", "heads", "reparam", "dec.0", "dec.2"]
LFWD = CHAIN + ["U = d2 A^T + q", "activations stored"]
LBWD = ["start", "weights, activations, S in LDS", "func.0 lane part (dd2, dA)", "dec.2 bwd", "dec.0 bwd", "latent", "heads bwd", "rho.2 bwd",
        "rho.0 bwd", "phi.2 bwd", "phi.0 bwd", "slab written"]


# the one-launch step: (stamp number, what ended there), in the order the kernel passes them
STEP = [(0, "start"), (44, "prologue: kernel arguments, p/q requests (waves 6, 7)"), (45, "prologue: phi.0, biases, func.2 requested"),
        (46, "prologue: cars requested"), (40, "prologue: operands requested"), (41, "prologue: biases arrived and stored"), (42, "prologue: W2, cars stored"),
        (1, "prologue: barrier"), (2, "tile start"), (22, "phi.0"), (23, "phi.2"), (24, "rho.0"), (25, "rho.2"),
        (26, "heads"), (27, "reparam"), (28, "dec.0"), (3, "dec.2"), (4, "U = d2 A^T + q"),
        (10, "car: inputs, func.0"), (11, "car: func.2 fwd"), (12, "car: func.4 + err"), (13, "car: func.4 bwd"),
        (14, "car: df2 W2"), (15, "car: barrier 1"), (16, "car: dW2 across waves"), (17, "car: barrier 2"),
        (18, "car: mask, column sums, df1 image"), (19, "car: per-lane sums"), (20, "car: barrier 3"), (5, "car: S accumulated (earlier rounds)"),
        (30, "operand requests, small sums parked, last round into S, S in LDS"), (31, "func.0 lane part (dA, dd2)"), (6, "barrier"), (32, "dec.2 bwd"),
        (33, "dec.0 bwd"), (34, "latent"), (35, "heads bwd"), (36, "rho.2 bwd"), (37, "rho.0 bwd"), (38, "phi.2 bwd"), (7, "phi.0 bwd"),
        (8, "slab, small sums, loc gradients")]


def main():
    B, N, g = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (4096, 7, 4)
    enc = Encoder(2, g, device="cuda:0", seed=1)
    rs = np.random.RandomState(0)
    x = torch.from_numpy(rs.uniform(0, 1, size=(B, N, 2)).astype(np.float32)).cuda()
    for _ in range(20):
        enc.learn(x, sync=False)
    torch.cuda.synchronize()
    lib = _lib.load()
    buf = (ctypes.c_longlong * (4 * 64))()
    lib.hhrl_debug_stamps.restype = ctypes.c_int
    assert lib.hhrl_debug_stamps(buf) == 0
    st = np.array(buf[:]).reshape(4, 64)
    if st[3, 0] != 0:                       # the one-launch form ran
        span = st[3, 8] - st[3, 0]
        print("vae_step: %d clocks" % span)
        prev = st[3, 0]
        for idx, label in STEP[1:]:
            d = st[3, idx] - prev
            prev = st[3, idx]
            print("  %-40s %8d  %5.1f%%" % (label, d, 100.0 * d / span))
        return
    for k, (name, labels) in enumerate((("vae_car_fwd_bwd", CAR), ("vae_lane_forward", LFWD), ("vae_lane_backward", LBWD))):
        t = st[k, :len(labels)]
        span = t[-1] - t[0]
        print("%s: %d clocks" % (name, span))
        for i in range(1, len(labels)):
            d = t[i] - t[i - 1]
            print("  %-34s %8d  %5.1f%%" % (labels[i], d, 100.0 * d / span))


    if os.environ.get("HHRL_STAMPS_RAW"):
        for k in range(3):
            print("raw", k, [int(v - st[k, 0]) for v in st[k]])


if __name__ == "__main__":
    main()
