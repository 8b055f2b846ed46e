"""The scenarios of the world-size-2 data-parallel test, written once and run two ways: by every rank of a
two-process group on its shard (tests/dp_worker.py), and by the test process alone on the whole global batch
(group=None, shard = everything).  Inputs are functions of fixed seeds, so all runs see the same global batch."""
import numpy as np
import torch

import gen

VAE_CASES = (("even", 64, 7, 4), ("uneven", 37, 5, 8))          # tag, global lanes, cars per lane, latent
TD3_UPDATES = 6                                                  # three plain and three policy steps: every launch
                                                                 # chain is run eagerly twice, captured, replayed


def _shard(n, rank, world, shard_bounds):
    return shard_bounds(n, rank, world) if world > 1 else (0, n)


def run_vae(res, group, rank, world, shard_bounds):
    from hhrl_vae_amd.algorithm import Encoder
    from hhrl_vae_amd.vae import vae_shapes
    for tag, B, N, g in VAE_CASES:
        init = gen.gen_state(vae_shapes(g), 100 + g)
        enc = Encoder(2, g, device="cuda:0", seed=5, group=group, static_shards=True)
        enc.load_state_dict(init)
        lo, hi = _shard(B, rank, world, shard_bounds)
        losses = []
        for step in range(3):
            x = torch.from_numpy(gen.gen_lanes(B, N, 300 + step)[lo:hi]).cuda()
            eps = torch.randn(B, g, generator=torch.Generator().manual_seed(40 + step))[lo:hi]
            # step 0 passes the global count, the others let the encoder all-reduce it (once: static shards)
            losses.append(enc.learn(x, eps=eps, global_count=B * N if step == 0 else None))
            if step == 0:
                for k, v in enc.grads().items():
                    res["vae_%s/grad0/%s" % (tag, k)] = v.numpy()
                for k, v in enc.state_dict().items():
                    res["vae_%s/param0/%s" % (tag, k)] = v.numpy()
        res["vae_%s/losses" % tag] = np.asarray(losses)
        for k, v in enc.state_dict().items():
            res["vae_%s/param/%s" % (tag, k)] = v.numpy()


def td3_net(kind, group):
    import hhrl_vae_amd.algorithm as A
    if kind == "worker":                 # the reference's widths: the fused three-layer kernels
        return A.WorkerTD3(gen.cav_cfg(cap=256, batch=33, a_sched=2, c_sched=3), device="cuda:0", seed=5, group=group)
    # the reference's widths (observations 13/13/25 over T=10, LSTM state 128, actors [400, 300], heads [512, 256]):
    # the one-launch recurrence kernels and every product kernel form run under the group
    cfg = gen.light_cfg(batch=16, cap=64, a_sched=2, c_sched=3)
    return A.HATD3Triple(cfg, device="cuda:0", seed=5, group=group, device_replay=True)


def run_td3(res, kind, group, rank, world, shard_bounds):
    net = td3_net(kind, group)
    n_rows = net.memory.shape[0]
    net.memory[:] = gen.gen_rows(n_rows, net.fam.row_width, 7)
    net.pointer = n_rows
    B = net.batch_size                                           # the GLOBAL minibatch; uneven over two ranks for 33
    lo, hi = _shard(B, rank, world, shard_bounds)
    rs = np.random.RandomState(11)
    n_act = len(net.actors)
    for it in range(TD3_UPDATES):
        idx = rs.randint(0, n_rows, size=B)
        noise = [rs.standard_normal((B, a)).astype(np.float32) for a in net.fam.a_dims]
        order = [(k + 2) % n_act for k in range(n_act)]     # one order throughout, so that its launch chain gets captured
        net.learn(indices=idx[lo:hi], noise=[n[lo:hi] for n in noise], order=order, global_batch=B)
        for k, v in net.critic.opt.g.state_dict().items():
            res["%s/it%d/grad/critic/%s" % (kind, it, k)] = v.numpy()
        res["%s/it%d/lr_critic" % (kind, it)] = np.float64(net.critic.opt.lr)
        if (it + 1) % net.policy_freq == 0:
            for name, a in zip(net.actor_names, net.actors):
                for k, v in a.opt.g.state_dict().items():
                    res["%s/it%d/grad/%s/%s" % (kind, it, name, k)] = v.numpy()
    for name, sd in net.state_dicts().items():
        for k, v in sd.items():
            res["%s/param/%s/%s" % (kind, name, k)] = v.numpy()
    for name, sd in net.target_state_dicts().items():
        for k, v in sd.items():
            res["%s/target/%s/%s" % (kind, name, k)] = v.numpy()
    res["%s/forms" % kind] = np.asarray(sorted(set(net.launch_forms().values())))


def run_noise_streams(res, group):
    """Noise the classes draw themselves: ranks share the seed (same initial weights) but not the stream."""
    from hhrl_vae_amd.algorithm import Encoder
    enc = Encoder(2, 4, device="cuda:0", seed=5, group=group)
    res["noise/vae_eps"] = enc._draw_eps(8, 4).cpu().numpy().copy()
    res["noise/vae_w"] = enc.state_dict()["encoder.rho.0.weight"].numpy()
    net = td3_net("worker", group)
    res["noise/td3_eps"] = net._np_rng.standard_normal(8)
    res["noise/td3_w"] = net.state_dicts()["critic"]["l1.weight"].numpy()
