#!/usr/bin/env python3
"""Shader-clock phase breakdown of the fused three-layer kernels (stamps build of the library).

Build first (CPU box): python hhrl-vae_amd/build.py --stamps
Run (GPU box):         python tools/mlp3_stamps.py M n
Prints the clocks between consecutive stamps of workgroup (0, 0): forward per tile = inputs in LDS, layer 1,
layer 2, layer 3 partials (+ hidden activations saved), outputs; backward per tile = inputs in LDS, tile done.
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["HHRL_HIP_LIB"] = os.path.join(ROOT, "hhrl-vae_amd", "lib", "libhhrl_hip_stamps.so")
sys.argv = [sys.argv[0]] + sys.argv[1:3] + ["3"]
sys.path.insert(0, os.path.join(ROOT, "tools"))
import mlp3_micro  # noqa: E402,F401  (runs the launches)
import torch  # noqa: E402

torch.cuda.synchronize()
lib = mlp3_micro.lib
buf = (C.c_longlong * 128)()
lib.hhrl_debug_mlp_stamps.restype = C.c_int
assert lib.hhrl_debug_mlp_stamps(buf) == 0
for k, name in ((0, "forward"), (1, "backward")):
    v = [buf[64 * k + i] for i in range(64)]
    n = max(i for i in range(64) if v[i] != 0) + 1
    d = [v[i] - v[i - 1] for i in range(1, n)]
    print("%s: %d stamps, total %d clocks; deltas: %s" % (name, n, v[n - 1] - v[0], d))
