"""The offline VAE training loop of the reference's VAE_train.py (train mode,
lines 46-69), as functions: load a state log, bucket lanes by car count, cut
buckets into chunks of <= 64 lanes, shuffle the chunks, and for each latent
size train a fresh model with ONE Adam step per chunk, then save it.

Sampling the log from SUMO (VAE_train.py:23-44) stays with the reference: it
needs the simulator.  The log format is the one that script writes:
``{str(n_cars): [[[x, v]] * n_cars, ...]}`` as JSON.
"""
from __future__ import annotations

import json
import os
import random
from typing import Callable, Dict, Iterable, List, Optional, Sequence

from .vae import Encoder


def read_state_log(path: str) -> Dict[str, list]:
    """utils.json_read as used at VAE_train.py:52."""
    with open(path, "r") as f:
        return json.load(f)


def chunk_dataset(dataset: Dict[str, list], batch_size: int = 64) -> List[list]:
    """VAE_train.py:55-56.  Lanes in one chunk share their car count (they
    come from one bucket); the last chunk of a bucket may be short; empty
    buckets vanish; bucket order is the dict's order."""
    chunks: List[list] = []
    for lanes in dataset.values():
        for i in range(0, len(lanes), batch_size):
            chunks.append(lanes[i: i + batch_size])
    return chunks


def shuffle_chunks(chunks: List[list], rng: Optional[random.Random] = None) -> List[list]:
    """VAE_train.py:57 (`random.shuffle`, unseeded there).  Pass a seeded
    `random.Random` for a reproducible order."""
    out = list(chunks)
    (rng or random).shuffle(out)
    return out


def prepare_chunks(chunks: Iterable, device="cuda") -> List:
    """Each chunk as an fp32 device tensor [B, N, 2], converted once.  The reference turns a chunk's nested
    lists into a tensor inside every `learn` call (algorithm.py:93), i.e. once per chunk per latent size;
    the conversion costs more host time than the whole training step takes on the GPU."""
    import numpy as np
    import torch
    return [torch.from_numpy(np.asarray(c, dtype=np.float32)).to(device) for c in chunks]


def train_one_pass(encoder: Encoder, chunks: Iterable, eps_list: Optional[Sequence] = None,
                   on_step: Optional[Callable[[int, float], None]] = None, sync_every: int = 1) -> List[float]:
    """VAE_train.py:65-68: one optimizer step per chunk, loss logged per step.

    sync_every > 1 keeps the loss on the device and fetches it in groups, so
    the device is not drained after every step (the reference's `loss.item()`
    at algorithm.py:107 forces that)."""
    losses: List[float] = []
    # device losses live in a ring of LOSS_RING slots, and two groups are in flight below
    sync_every = min(sync_every, (encoder.LOSS_RING - 1) // 2)
    if sync_every <= 1:
        for step, data in enumerate(chunks):
            loss = encoder.learn(data, eps=None if eps_list is None else eps_list[step])
            losses.append(loss)
            if on_step:
                on_step(step, loss)
        return losses

    def fetch(ts):
        import torch
        # one device-to-host copy for the group (a .item() per loss is a synchronous copy each: 15 us per step)
        for v in torch.stack(list(ts)).tolist():
            losses.append(float(v))
            if on_step:
                on_step(len(losses) - 1, losses[-1])

    # Groups of sync_every steps; the losses of a group are read only after the NEXT group has been enqueued, so the
    # device never runs dry while the host reads them (reading right after a group drains the queue: 66 instead
    # of 50 us per step on the 64-lane chunks of the reference's job).  A group is one library call where the
    # encoder offers it (learn_many), else one learn() per chunk.
    chunks = list(chunks)
    many = getattr(encoder, "learn_many", None)
    prev = None
    for lo in range(0, len(chunks), sync_every):
        part = chunks[lo:lo + sync_every]
        eps = None if eps_list is None else eps_list[lo:lo + len(part)]
        if many is not None:
            cur = many(part, eps)
        else:
            cur = [encoder.learn(c, eps=None if eps is None else eps[i], sync=False) for i, c in enumerate(part)]
        if prev is not None:
            fetch(prev)
        prev = cur
    if prev is not None:
        fetch(prev)
    return losses


def train_all(dataset: Dict[str, list], goal_dims: Sequence[int] = (2, 4, 8, 16), batch_size: int = 64,
              save_dir: Optional[str] = None, shuffle_rng: Optional[random.Random] = None,
              on_step: Optional[Callable[[int, int, float], None]] = None, sync_every: int = 1):
    """VAE_train.py:55-69: the same shuffled chunk list is replayed once for
    every latent size on a freshly initialised model; each model is saved as
    `<save_dir>/encoder_dim_<g>` (+ `_enc`, `_dec`, `_encoder_optimizer`)."""
    shuffled = shuffle_chunks(chunk_dataset(dataset, batch_size), shuffle_rng)
    shuffled = prepare_chunks(shuffled)              # the same chunks are replayed for every latent size
    out = {}
    for g in goal_dims:
        enc = Encoder(2, g)
        cb = (lambda s, l, g=g: on_step(g, s, l)) if on_step else None
        losses = train_one_pass(enc, shuffled, on_step=cb, sync_every=sync_every)
        if save_dir is not None:
            os.makedirs(save_dir, exist_ok=True)
            enc.save(os.path.join(save_dir, "encoder_dim_" + str(g)))
        out[g] = (enc, losses)
    return out
