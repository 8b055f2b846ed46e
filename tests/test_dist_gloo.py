"""World-size-2 data parallelism on CPU (gloo): the sharding and scaling rules
of the N>1 path.  Each rank computes its shard's gradients with the CPU oracle
(the HIP kernels need a GPU), exchanges them through the product's all-reduce
helper, and must land on the single-process result for the global batch."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "golden")):
        sys.path.insert(0, p)
    import torch.distributed as dist
    import gen
    import hhrl_oracle as O
    from hhrl_vae_amd import dist as hdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    group = dist.group.WORLD
    res = {}
    # ---- VAE: mean-normalised reconstruction + summed KL
    g, B, N = 4, 10, 5                                  # 10 lanes -> shards of 5 and 5; also try uneven below
    init = {k: torch.from_numpy(v) for k, v in gen.gen_state(O.vae_param_shapes(g), 5).items()}
    x = torch.from_numpy(gen.gen_lanes(B, N, 6))
    eps = torch.randn(B, g, generator=torch.Generator().manual_seed(7))
    for tag, nb in (("even", 10), ("uneven", 9)):
        lo, hi = hdist.shard_bounds(nb, rank, world)
        vae = O.VAE(init, g)
        cnt = hdist.global_count((hi - lo) * N, group)
        assert cnt == nb * N
        loss, grads = vae.loss_and_grads(x[lo:hi], eps[lo:hi], recon_count=cnt)
        flat = torch.cat([grads[k].reshape(-1) for k in vae.names])
        hdist.allreduce_sum_(flat, group)
        lt = hdist.allreduce_sum_(loss.clone().reshape(1), group)
        pieces = torch.split(flat, [grads[k].numel() for k in vae.names])
        vae.opt.step([pc.reshape(grads[k].shape) for pc, k in zip(pieces, vae.names)])
        res["vae_%s_grad" % tag] = flat.numpy()
        res["vae_%s_loss" % tag] = lt.numpy()
        res["vae_%s_param" % tag] = torch.cat([vae.p[k].detach().reshape(-1) for k in vae.names]).numpy()
    # ---- TD3 critic: plain means over the batch
    cfg = gen.cav_cfg(o=5, g=3, hidden=(16, 12), batch=8, cap=48)
    fam = O.family_from_cfg("worker", cfg)
    t = lambda d: {k: torch.from_numpy(v) for k, v in d.items()}
    net = O.TD3(fam, [t(gen.gen_state(O.actor_param_shapes(fam, 0), 1))], t(gen.gen_state(O.critic_param_shapes(fam), 2)))
    rows = gen.gen_rows(8, fam.row_width, 3)
    noise = torch.randn(8, 1, generator=torch.Generator().manual_seed(4))
    lo, hi = hdist.shard_bounds(8, rank, world)
    b = O.split_rows(fam, rows[lo:hi])
    y = net.target_q(b, [noise[lo:hi]])
    q1, q2 = O.critic_forward(net.critic, fam, b.obs, torch.cat(b.act, 1), b.goal)
    inv_n = 1.0 / hdist.global_count(hi - lo, group)
    closs = inv_n * (((q1 - y) ** 2).sum() + ((q2 - y) ** 2).sum())
    cg = torch.autograd.grad(closs, [net.critic[k] for k in net.critic_names])
    flat = torch.cat([x_.reshape(-1) for x_ in cg])
    hdist.allreduce_sum_(flat, group)
    res["td3_grad"] = flat.numpy()
    # ---- the actor order of a multi-actor update: drawn per rank, rank 0's taken by all
    mine = np.random.RandomState(100 + rank).permutation(3)
    res["order_mine"] = mine
    res["order_agreed"] = np.asarray(hdist.broadcast_ints(mine, group))
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **res)
    dist.barrier()
    dist.destroy_process_group()


def test_world2_matches_single_process(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import gen
    import hhrl_oracle as O
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(os.path.join(str(tmp_path), "rank0.npz"))
    r1 = np.load(os.path.join(str(tmp_path), "rank1.npz"))
    for k in r0.files:                                     # replicas stay identical
        if k != "order_mine":
            assert np.array_equal(r0[k], r1[k]), k
    assert not np.array_equal(r0["order_mine"], r1["order_mine"])                 # the ranks drew different orders
    assert np.array_equal(r1["order_agreed"], r0["order_mine"]) and sorted(r1["order_agreed"]) == [0, 1, 2]
    g, N = 4, 5
    init = {k: torch.from_numpy(v) for k, v in gen.gen_state(O.vae_param_shapes(g), 5).items()}
    x = torch.from_numpy(gen.gen_lanes(10, N, 6))
    eps = torch.randn(10, g, generator=torch.Generator().manual_seed(7))
    for tag, nb in (("even", 10), ("uneven", 9)):
        vae = O.VAE(init, g)
        loss = vae.learn(x[:nb].numpy(), eps[:nb])
        flat = torch.cat([vae.grads[k].reshape(-1) for k in vae.names]).numpy()
        np.testing.assert_allclose(r0["vae_%s_grad" % tag], flat, rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(r0["vae_%s_loss" % tag], [loss], rtol=1e-5)
        np.testing.assert_allclose(r0["vae_%s_param" % tag], torch.cat([vae.p[k].detach().reshape(-1) for k in vae.names]).numpy(),
                                   rtol=1e-4, atol=2e-6)
    cfg = gen.cav_cfg(o=5, g=3, hidden=(16, 12), batch=8, cap=48)
    fam = O.family_from_cfg("worker", cfg)
    t = lambda d: {k: torch.from_numpy(v) for k, v in d.items()}
    net = O.TD3(fam, [t(gen.gen_state(O.actor_param_shapes(fam, 0), 1))], t(gen.gen_state(O.critic_param_shapes(fam), 2)))
    rows = gen.gen_rows(8, fam.row_width, 3)
    noise = torch.randn(8, 1, generator=torch.Generator().manual_seed(4))
    out = net.learn(O.split_rows(fam, rows), [noise])
    flat = torch.cat([out["critic_grads"][k].reshape(-1) for k in net.critic_names]).numpy()
    np.testing.assert_allclose(r0["td3_grad"], flat, rtol=2e-5, atol=1e-7)
