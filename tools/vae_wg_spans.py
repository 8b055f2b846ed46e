#!/usr/bin/env python3
"""When the workgroups of the one-launch VAE step start and end (diagnostic build).

Build first (CPU box): python hhrl-vae_amd/build.py --stamps
Run (GPU box):         python tools/vae_wg_spans.py [B N g]

Thread 0 of every workgroup records the constant-rate counter (100 MHz) and the shader clock when it starts and
when it ends; the script prints how the starts and ends spread and the shader clock rate the spans imply."""
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


def main():
    B, N, g = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (4096, 7, 4)
    enc = Encoder(2, g, device="cuda:0", seed=1)
    rs = np.random.RandomState(0)
    x = torch.from_numpy(rs.uniform(0, 1, size=(B, N, 2)).astype(np.float32)).cuda()
    lib = _lib.load()
    lib.hhrl_debug_wg_spans.restype = ctypes.c_int
    for trial in range(3):
        for _ in range(200):
            enc.learn(x, sync=False)
        torch.cuda.synchronize()
        buf = (ctypes.c_longlong * (256 * 4))()
        assert lib.hhrl_debug_wg_spans(buf) == 0
        a = np.array(buf[:]).reshape(256, 4)
        n = min(256, (B + 15) // 16)
        a = a[:n]
        t0 = a[:, 0].min()
        st, en = (a[:, 0] - t0) * 10.0, (a[:, 1] - t0) * 10.0          # ns at 100 MHz
        dur = en - st
        clk = (a[:, 3] - a[:, 2]) / np.maximum(dur, 1)                   # shader clocks per ns
        print("trial %d: %d workgroups" % (trial, n))
        print("  start  ns: min %6.0f  median %6.0f  p90 %6.0f  max %6.0f" % (st.min(), np.median(st), np.percentile(st, 90), st.max()))
        print("  end    ns: min %6.0f  median %6.0f  p90 %6.0f  max %6.0f" % (en.min(), np.median(en), np.percentile(en, 90), en.max()))
        print("  span   ns: min %6.0f  median %6.0f  p90 %6.0f  max %6.0f" % (dur.min(), np.median(dur), np.percentile(dur, 90), dur.max()))
        print("  shader clock: median %.3f GHz;  workgroup 0: start %.0f end %.0f ns, %d clocks" % (np.median(clk), st[0], en[0], a[0, 3] - a[0, 2]))
        order = np.argsort(en)[-5:]
        print("  last to end: " + ", ".join("wg %d (start %.0f, end %.0f)" % (i, st[i], en[i]) for i in order))


if __name__ == "__main__":
    main()
