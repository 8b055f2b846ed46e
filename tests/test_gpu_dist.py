"""The data-parallel code path on a real GPU with a one-rank RCCL group: the
process group, the flat gradient(+loss) all-reduce and the separate Adam
launch must reproduce the single-device step (a one-rank SUM is the identity):
bit for bit for the VAE, apart from the order of float atomics for the TD3 updates.  The two-rank arithmetic is covered on CPU by test_dist_gloo."""
import os

import numpy as np
import pytest
import torch

import gen
import hhrl_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def group():
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29571")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist.group.WORLD
    if created:
        torch.cuda.synchronize()
        dist.destroy_process_group()


def test_vae_one_rank_group_matches_single_device(group):
    from hhrl_vae_amd.algorithm import Encoder
    g, B, N = 4, 257, 7
    init = {k: torch.from_numpy(v) for k, v in gen.gen_state(O.vae_param_shapes(g), 11).items()}
    a, b = Encoder(2, g, device="cuda:0"), Encoder(2, g, device="cuda:0", group=group)
    a.load_state_dict(init)
    b.load_state_dict(init)
    for step in range(3):
        x = torch.from_numpy(gen.gen_lanes(B, N, 40 + step)).cuda()
        eps = torch.randn(B, g, generator=torch.Generator().manual_seed(step)).cuda()
        la = a.learn(x, eps=eps)
        lb = b.learn(x, eps=eps)
        assert la == lb
    sa, sb = a.state_dict(), b.state_dict()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    assert a.opt.step_count == b.opt.step_count == 3


def test_td3_one_rank_group_matches_single_device(group):
    from hhrl_vae_amd.algorithm import WorkerTD3
    cfg = gen.cav_cfg(cap=256)
    a, b = WorkerTD3(cfg, device="cuda:0", seed=5), WorkerTD3(cfg, device="cuda:0", seed=5, group=group)
    rows = np.random.RandomState(2).uniform(0, 1, size=(256, a.fam.row_width))
    for net in (a, b):
        net.memory[:] = rows
        net.pointer = 256
    rs = np.random.RandomState(3)
    for step in range(4):
        idx = rs.randint(0, 256, size=a.batch_size)
        noise = [rs.standard_normal((a.batch_size, d)).astype(np.float32) for d in a.fam.a_dims]
        a.learn(indices=idx, noise=noise)
        b.learn(indices=idx, noise=noise)
    # equal apart from the order in which the four workgroups of a network add their weight gradients (float atomics)
    from test_gpu_kernels import scaled_close
    for (na, sa), (nb, sb) in zip(a.state_dicts().items(), b.state_dicts().items()):
        for k in sa:
            scaled_close(na + "/" + k, sb[k].cpu().numpy(), sa[k].cpu().numpy(), rtol=1e-5, atol_rel=1e-6)


def test_multi_actor_group_update_draws_one_order(group):
    """HATD3Triple with a process group and no injected actor order: the order is drawn from numpy's global
    generator as in the reference (algorithm.py:809) and passed through the group's broadcast, so that every
    rank steps (and all-reduces) the actors in the same sequence; with one rank the result is the single-device
    update under the same numpy seed, bit for bit apart from the float atomics of the weight-gradient kernels."""
    from hhrl_vae_amd.algorithm import HATD3Triple
    from test_gpu_kernels import scaled_close
    cfg = gen.light_cfg(o=(5, 6, 7), T=4, s=8, a=(1, 2, 3), actor_h=(12, 10), critic_h=(14, 12), batch=8, cap=32)
    a, b = HATD3Triple(cfg, device="cuda:0", seed=5), HATD3Triple(cfg, device="cuda:0", seed=5, group=group)
    rows = np.random.RandomState(2).uniform(0, 1, size=(32, a.fam.row_width))
    for net in (a, b):
        net.memory[:] = rows
        net.pointer = 32
    rs = np.random.RandomState(3)
    seen = []
    for step in range(6):
        idx = rs.randint(0, 32, size=8)
        noise = [rs.standard_normal((8, d)).astype(np.float32) for d in a.fam.a_dims]
        for net in (a, b):
            np.random.seed(50 + step)
            net.learn(None, indices=idx, noise=noise)
        np.random.seed(50 + step)
        seen.append(tuple(np.random.permutation(3)))
    assert len(set(seen)) > 1                                  # several different orders were exercised
    for (na, sa), (nb, sb) in zip(a.state_dicts().items(), b.state_dicts().items()):
        for k in sa:
            scaled_close(na + "/" + k, sb[k].cpu().numpy(), sa[k].cpu().numpy(), rtol=1e-5, atol_rel=1e-6)


def _pair(make, n_rows, seed=2):
    a, b = make(None), make("group")
    rows = np.random.RandomState(seed).uniform(0, 1, size=(n_rows, a.fam.row_width))
    for net in (a, b):
        net.memory[:n_rows] = rows
        net.pointer = n_rows
    return a, b


@pytest.mark.parametrize("in_graph", [False, True])
def test_worker_reference_widths_every_chain_replayed_under_rccl(group, in_graph):
    """WorkerTD3 at the reference's widths under the one-rank RCCL group for ten updates: both launch chains
    (plain and policy step) pass their eager runs, are captured -- as `graph, all-reduce, graph, ...` segments, or
    (in_graph) as one graph that holds the collectives -- and are replayed several times; equal to the
    single-device learner apart from the order of the weight-gradient atomics (four workgroups per network)."""
    from hhrl_vae_amd.algorithm import WorkerTD3
    from test_gpu_kernels import scaled_close
    cfg = gen.cav_cfg(cap=256)

    def make(g):
        net = WorkerTD3(cfg, device="cuda:0", seed=5, group=group if g else None)
        if g:
            net.graph_collectives = in_graph
        return net
    a, b = _pair(make, 256)
    rs = np.random.RandomState(3)
    for step in range(10):
        idx = rs.randint(0, 256, size=a.batch_size)
        noise = [rs.standard_normal((a.batch_size, d)).astype(np.float32) for d in a.fam.a_dims]
        a.learn(indices=idx, noise=noise)
        b.learn(indices=idx, noise=noise)
    forms = b.launch_forms()
    assert len(forms) == 2 and set(forms.values()) == {"graph"}, forms
    if not in_graph:
        assert sorted(len(s) for s in b._seg_graphs.values()) == [2, 3]       # critic | critic, actor | rest
    for get in ("state_dicts", "target_state_dicts"):
        for (na, sa), (nb, sb) in zip(getattr(a, get)().items(), getattr(b, get)().items()):
            for k in sa:
                scaled_close("%s/%s/%s" % (get, na, k), sb[k].cpu().numpy(), sa[k].cpu().numpy(), rtol=1e-5, atol_rel=1e-6)


@pytest.mark.parametrize("in_graph", [False, True])
def test_triple_reference_widths_every_chain_replayed_under_rccl(group, in_graph):
    """HATD3Triple at the reference's widths (13/13/25 x T=10, state 128, [400, 300] actors, [512, 256] heads) under
    the one-rank RCCL group, one actor order throughout, ten updates: the plain chain (2 segments) and the policy
    chain (5 segments: critic, three actors, rest) are each captured and replayed three times, with the one-launch
    recurrence kernels inside.  Against the single-device learner: equal apart from the float atomics of the
    weight-gradient products."""
    from hhrl_vae_amd.algorithm import HATD3Triple
    from test_gpu_kernels import scaled_close
    cfg = gen.light_cfg(cap=128)

    def make(g):
        net = HATD3Triple(cfg, device="cuda:0", seed=5, group=group if g else None)
        if g:
            net.graph_collectives = in_graph
        return net
    a, b = _pair(make, 128)
    rs = np.random.RandomState(3)
    for step in range(10):
        idx = rs.randint(0, 128, size=a.batch_size)
        noise = [rs.standard_normal((a.batch_size, d)).astype(np.float32) for d in a.fam.a_dims]
        a.learn(None, indices=idx, noise=noise, order=(2, 0, 1))
        b.learn(None, indices=idx, noise=noise, order=(2, 0, 1))
    forms = b.launch_forms()
    assert len(forms) == 2 and set(forms.values()) == {"graph"}, forms
    if not in_graph:
        assert sorted(len(s) for s in b._seg_graphs.values()) == [2, 5]
    for get in ("state_dicts", "target_state_dicts"):
        for (na, sa), (nb, sb) in zip(getattr(a, get)().items(), getattr(b, get)().items()):
            for k in sa:
                scaled_close("%s/%s/%s" % (get, na, k), sb[k].cpu().numpy(), sa[k].cpu().numpy(), rtol=1e-5, atol_rel=1e-6)
