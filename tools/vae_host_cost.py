"""Host cost of one Encoder.learn call against the GPU time of the step it launches.
usage: python tools/vae_host_cost.py [B N g]   (GPU box)
Prints the time the Python loop needs to ENQUEUE a step (measured before the final synchronize) and the
time per step including the synchronize: when the two agree the loop is bound by the host."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from hhrl_vae_amd.algorithm import Encoder


def main():
    B, N, g = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (4096, 7, 4)
    enc = Encoder(2, g, device="cuda:0", seed=1)
    x = torch.rand(B, N, 2, device="cuda:0")
    for _ in range(50):
        enc.learn(x, sync=False)
    torch.cuda.synchronize()
    K = 2000
    t0 = time.perf_counter()
    for _ in range(K):
        enc.learn(x, sync=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("B=%d N=%d g=%d: enqueue %.1f us/step, with synchronize %.1f us/step, queue drained in %.1f us after the loop"
          % (B, N, g, 1e6 * (t1 - t0) / K, 1e6 * (t2 - t0) / K, 1e6 * (t2 - t1)))


if __name__ == "__main__":
    main()
