#!/usr/bin/env python3
"""The reference's own offline workload (VAE_train.py:52-69): state-log lanes bucketed by car count,
chunks of 64, one pass per latent size -- synthetic log in the on-disk format (N = 1..40, 512 lanes each).
Times the HIP path and, on a sample of the chunks, the CPU oracle."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "golden")]
import numpy as np, torch
import gen, hhrl_oracle as O
from hhrl_vae_amd.algorithm import Encoder
from hhrl_vae_amd.vae_train import chunk_dataset, prepare_chunks, shuffle_chunks, train_one_pass

rng = np.random.default_rng(3407)
dataset = {}
for N in range(1, 41):
    x = rng.uniform(0, 1, size=(512, N, 1)); v = np.clip(rng.uniform(0, 1.08, size=(512, N, 1)), 0, 1.08)
    dataset[str(N)] = np.concatenate([x, v], -1).tolist()
chunks = shuffle_chunks(chunk_dataset(dataset, 64), random.Random(3407))
print("chunks:", len(chunks), "lanes:", sum(len(c) for c in chunks))
for g in (2, 4, 8, 16):
    enc = Encoder(2, g)
    train_one_pass(enc, chunks[:20], sync_every=16)
    torch.cuda.synchronize()
    for se in (1, 64):
        t0 = time.perf_counter()
        losses = train_one_pass(enc, chunks, sync_every=se)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("g=%d sync_every=%d: %.1f ms per pass, %.1f us/step, final loss %.4f" % (g, se, 1e3 * dt, 1e6 * dt / len(chunks), losses[-1]))
    # pre-converted device tensors: the kernels alone
    t0 = time.perf_counter()
    dev_chunks = prepare_chunks(chunks)
    torch.cuda.synchronize()
    print("g=%d prepare_chunks: %.1f ms once (%.1f us per chunk)" % (g, 1e3 * (time.perf_counter() - t0), 1e6 * (time.perf_counter() - t0) / len(chunks)))
    t0 = time.perf_counter()
    train_one_pass(enc, dev_chunks, sync_every=64)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("g=%d device-resident chunks, groups of 64 steps per library call: %.1f us/step" % (g, 1e6 * dt / len(chunks)))
    t0 = time.perf_counter()
    pend = [enc.learn(c, sync=False) for c in dev_chunks]
    last = pend[-1].item()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("g=%d device-resident chunks, one learn() per chunk: %.1f us/step" % (g, 1e6 * dt / len(chunks)))
shapes = O.vae_param_shapes(4)
vae = O.VAE({k: torch.from_numpy(v) for k, v in gen.gen_state(shapes, 1).items()}, 4)
sample = chunks[:60]
t0 = time.perf_counter()
for c in sample:
    vae.learn(c, torch.randn(len(c), 4))
dt = time.perf_counter() - t0
print("CPU oracle (g=4, %d threads): %.1f us/step" % (torch.get_num_threads(), 1e6 * dt / len(sample)))
