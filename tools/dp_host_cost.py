#!/usr/bin/env python3
"""Host-side cost of one data-parallel VAE step (one rank, RCCL group): microseconds the CPU spends
enqueueing each piece, without waiting for the GPU.  Run on a GPU box: python tools/dp_host_cost.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")

import numpy as np
import torch
import torch.distributed as dist


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    from hhrl_vae_amd import dist as hdist
    from hhrl_vae_amd.algorithm import Encoder
    B, N, g = 4096, 7, 4
    enc = Encoder(2, g, device="cuda:0", seed=1, group=dist.group.WORLD)
    x = torch.from_numpy(np.random.RandomState(0).uniform(0, 1, size=(B, N, 2)).astype(np.float32)).cuda()
    gc = B * N
    for _ in range(50):
        enc.learn(x, sync=False, global_count=gc)
    torch.cuda.synchronize()
    n = 300
    acc = {"loss_and_grads": 0.0, "allreduce": 0.0, "loss copy": 0.0, "adam": 0.0}
    gb = enc.opt.g
    t_all = time.perf_counter()
    for _ in range(n):
        t0 = time.perf_counter()
        enc.loss_and_grads(x, None, recon_count=gc, _loss_to_tail=True)
        t1 = time.perf_counter()
        hdist.allreduce_sum_(gb.full, enc.group)
        t2 = time.perf_counter()
        enc._loss.copy_(gb.full[enc.layout.size:enc.layout.size + 1])
        t3 = time.perf_counter()
        enc.opt.step()
        t4 = time.perf_counter()
        acc["loss_and_grads"] += t1 - t0
        acc["allreduce"] += t2 - t1
        acc["loss copy"] += t3 - t2
        acc["adam"] += t4 - t3
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t_all
    for k, v in acc.items():
        print("%-16s %7.1f us host" % (k, 1e6 * v / n))
    print("step wall        %7.1f us" % (1e6 * t_all / n))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
