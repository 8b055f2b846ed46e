"""TD3 family updates on the GPU (through the C ABI) replayed against the
golden vectors recorded from the reference: same initial weights, replay rows,
sampled indices, smoothing noise and actor order.  Bar: fp32, rtol 1e-4 (plus
an absolute floor tied to each tensor's scale, see test_gpu_kernels)."""
import os

import numpy as np
import pytest
import torch

import gen
import hhrl_oracle as O
from golden_util import adam_path_close, assert_close, load_case, sub, td3_inputs
from test_gpu_kernels import scaled_close

pytestmark = pytest.mark.gpu


def build(kind, task, cfg):
    import hhrl_vae_amd.algorithm as A
    if kind == "single":
        return A.TD3Single(cfg, task)
    if kind == "double":
        return A.HATD3Double(cfg, task)
    if kind == "tp":
        return A.HATD3(cfg)
    if kind == "triple":
        return A.HATD3Triple(cfg)
    if kind == "manager":
        return A.ManagerTD3(cfg)
    return A.WorkerTD3(cfg)


def fill_memory(net, kind, rows):
    n = rows.shape[0]
    if kind == "manager":
        od, ad = net.obs_dim, net.a_dim
        b = net.memory.buffer
        b["obs"][:n], b["action"][:n] = rows[:, :od], rows[:, od: od + ad]
        b["reward"][:n], b["next_obs"][:n] = rows[:, od + ad: od + ad + 1], rows[:, od + ad + 1:]
        net.memory.pointer = n
    else:
        net.memory[:n] = rows
        net.pointer = n


def close_dict(name, got, want, digest, atol_rel=2e-6):
    assert set(want) <= set(got), (name, sorted(set(want) - set(got)))
    for k, w in want.items():
        g = got[k].numpy() if isinstance(got[k], torch.Tensor) else np.asarray(got[k])
        if not digest:
            scaled_close(name + "/" + k, g, w, atol_rel=atol_rel)
            continue
        d = gen.digest(g)
        mean_abs = w[1] / max(g.size, 1)
        assert_close(name + "/" + k + "[sum]", d[0], w[0], rtol=1e-4, atol=atol_rel * 10 * max(w[1], 1e-30))
        assert_close(name + "/" + k + "[abs]", d[1], w[1], rtol=1e-4, atol=0)
        assert_close(name + "/" + k + "[samples]", d[2:], w[2:], rtol=1e-4, atol=atol_rel * 20 * max(mean_abs, 1e-30))


@pytest.mark.parametrize("case", list(gen.TD3_CASES))
def test_td3_golden(case):
    kind, task, mkcfg, n_rows, n_calls, mode = gen.TD3_CASES[case]
    fx = load_case("td3_" + case)
    cfg = mkcfg()
    net = build(kind, task, cfg)
    fam = net.fam
    names = net.actor_names
    ofam = O.family_from_cfg(kind, mkcfg(), task)
    shapes = {n: O.actor_param_shapes(ofam, k) for k, n in enumerate(names)}
    shapes["critic"] = O.critic_param_shapes(ofam)
    init, rows = td3_inputs(case, fx, shapes)
    assert rows.shape[1] == fam.row_width == int(fx["row_width"])
    net.load_state_dicts(init)
    fill_memory(net, kind, rows)
    dg = mode == "digest"
    for call in range(n_calls):
        pre = "call%d" % call
        noise = [fx[pre + "/eps%d" % j] for j in range(len(names))]
        net.learn(indices=fx[pre + "/indices"], noise=noise, order=fx[pre + "/order"])
        scaled_close(pre + "/mse", net.last_critic_loss.cpu().numpy(), fx[pre + "/mse"])
        close_dict(pre + "/grad/critic", net.critic.opt.g.state_dict(), sub(fx, pre + "/grad/critic"), dg, atol_rel=5e-6)
        close_dict(pre + "/param/critic", net.critic.params.state_dict(), sub(fx, pre + "/param/critic"), dg)
        if bool(fx[pre + "/policy"]):
            for k, n in enumerate(names):
                close_dict(pre + "/grad/" + n, net.actors[k].opt.g.state_dict(), sub(fx, pre + "/grad/" + n), dg, atol_rel=5e-6)
                close_dict(pre + "/param/" + n, net.actors[k].params.state_dict(), sub(fx, pre + "/param/" + n), dg)
                close_dict(pre + "/target/" + n, net.actor_targets[k].params.state_dict(), sub(fx, pre + "/target/" + n), dg)
            close_dict(pre + "/target/critic", net.critic_target.params.state_dict(), sub(fx, pre + "/target/critic"), dg)
        assert_close(pre + "/lr_critic", net.critic.opt.lr, fx[pre + "/lr_critic"], rtol=1e-12, atol=0)
        assert_close(pre + "/lr_actor", net.actors[0].opt.lr, fx[pre + "/lr_actor"], rtol=1e-12, atol=0)
    for k, n in enumerate(names + ["critic"]):
        opt = net.critic.opt if n == "critic" else net.actors[k].opt
        close_dict("exp_avg/" + n, opt.m.state_dict(), sub(fx, "final/exp_avg/" + n), dg, atol_rel=5e-6)
        close_dict("exp_avg_sq/" + n, opt.v.state_dict(), sub(fx, "final/exp_avg_sq/" + n), dg, atol_rel=1e-5)
    if not dg:
        fns = {"single": ["choose_action"], "manager": ["choose_action"], "worker": ["choose_action"],
               "double": ["choose_light_action", "choose_goal"], "tp": ["choose_time_action", "choose_phase_action"],
               "triple": ["choose_time_action", "choose_phase_action", "choose_goal"]}[kind]
        for k, n in enumerate(names):
            args = [fx["choose/%s/o" % n]] + ([fx["choose/%s/g" % n]] if kind == "worker" else [])
            out = getattr(net, fns[k])(*args)
            assert out.ndim == 1
            scaled_close("choose/" + n, out, fx["choose/%s/out" % n], atol_rel=5e-6)
        if "policy/out" in fx:
            scaled_close("policy", net.policy(fx["policy/o"], fx["policy/g"]), fx["policy/out"], atol_rel=5e-6)


def test_policy_calls_staged_and_in_place_agree(monkeypatch):
    """choose_* / policy with the calls' pinned blocks used in place by the kernels (default) and staged through
    device buffers (HHRL_ZERO_COPY=0): the same bits, for a fused three-layer actor and for LSTM actors, whose
    batch-1 evaluation turns into a graph replay on the third call."""
    import hhrl_vae_amd.algorithm as A
    rs = np.random.RandomState(6)
    o_w, g_w = rs.uniform(0, 1, size=(5, 11)).astype(np.float32), rs.uniform(0, 1, size=(5, 6)).astype(np.float32)
    o_t = [rs.uniform(0, 1, size=(130,)).astype(np.float32) for _ in range(4)]
    o_g = rs.uniform(0, 1, size=(8, 250)).astype(np.float32)

    def run():
        w, t = A.WorkerTD3(gen.cav_cfg(cap=64), seed=3), A.HATD3Triple(gen.light_cfg(cap=64), seed=3)
        out = [w.choose_action(o_w[0], g_w[0]), w.policy(o_w, g_w)]
        out += [t.choose_time_action(o) for o in o_t]                       # eager, eager, capture + replay, replay
        out += [t.choose_phase_action(o_t[0]), t.choose_goal_batch(o_g)]
        return (w, t), out

    (w, t), ra = run()
    assert all(io["zc"] for net in (w, t) for io in net._policy_io.values())
    monkeypatch.setenv("HHRL_ZERO_COPY", "0")
    (w2, t2), rb = run()
    assert not any(io["zc"] for net in (w2, t2) for io in net._policy_io.values())
    for x, y in zip(ra, rb):
        assert np.array_equal(np.asarray(x), np.asarray(y))


@pytest.mark.parametrize("kind,batch", [("worker", 4096), ("triple", 1024)])
def test_td3_benched_batch_sizes_vs_oracle(kind, batch):
    """The batch sizes bench.py times beside the reference's 64 -- WorkerTD3 at 4096 rows and HATD3Triple at
    1024 -- against the CPU oracle on identical weights, rows, indices, noise and actor order: two calls, the
    second a policy step.  At these sizes other kernels run than at batch 64 (device-resident replay and its
    gather, the large-tile GEMM, weight gradients split over the rows with float atomics).  Bars: loss and
    gradients rtol 1e-4 with a floor of 1e-5 of the tensor's largest entry (sums over up to 4096 rows in a
    different order), parameters by golden_util.adam_path_close from the oracle's gradients."""
    import hhrl_vae_amd.algorithm as A
    cfg = gen.cav_cfg(cap=8192, batch=batch) if kind == "worker" else gen.light_cfg(cap=2048, batch=batch)
    net = A.WorkerTD3(cfg) if kind == "worker" else A.HATD3Triple(cfg)
    assert net.device_replay
    ofam = O.family_from_cfg(kind, cfg)
    names = net.actor_names
    t = lambda d: {k: torch.from_numpy(v) for k, v in d.items()}
    a_init = [gen.gen_state(O.actor_param_shapes(ofam, k), 400 + k) for k in range(len(names))]
    c_init = gen.gen_state(O.critic_param_shapes(ofam), 410)
    oracle = O.TD3(ofam, [t(a) for a in a_init], t(c_init))
    net.load_state_dicts(dict(zip(names, a_init), critic=c_init))
    n_rows = net.memory.shape[0]
    rows = gen.gen_rows(n_rows, net.fam.row_width, 411)
    net.memory[:] = rows
    net.pointer = n_rows
    rs = np.random.RandomState(412)
    glog = {n: [] for n in names + ["critic"]}
    lrlog = {n: [] for n in names + ["critic"]}
    for call in range(2):
        idx = rs.randint(0, n_rows, size=batch)
        noise = [rs.standard_normal((batch, a)).astype(np.float32) for a in net.fam.a_dims]
        order = list(rs.permutation(len(names)))
        lr_c, lr_a = oracle.critic_opt.lr, [o.lr for o in oracle.actor_opts]
        out = oracle.learn(O.split_rows(ofam, rows[idx]), [torch.from_numpy(n) for n in noise], order)
        net.learn(indices=idx, noise=noise, order=order)
        pre = "%s_b%d/call%d" % (kind, batch, call)
        mse = net.last_critic_loss.cpu().numpy()
        assert_close(pre + "/critic_loss", float(mse[0] + mse[1]), out["critic_loss"], rtol=1e-4, atol=0)
        for k, w in out["critic_grads"].items():
            scaled_close(pre + "/grad/critic/" + k, net.critic.opt.g[k].cpu().numpy(), w.numpy(), rtol=1e-4, atol_rel=1e-5)
        glog["critic"].append({k: v.numpy() for k, v in out["critic_grads"].items()})
        lrlog["critic"].append(lr_c)
        # the last actor stepped still holds its gradient (an earlier one's buffer is intact too: zeroed per actor)
        for i, g in out.get("actor_grads", {}).items():
            for k, w in g.items():
                scaled_close("%s/grad/%s/%s" % (pre, names[i], k), net.actors[i].opt.g[k].cpu().numpy(), w.numpy(),
                             rtol=1e-4, atol_rel=2e-5)
            glog[names[i]].append({k: v.numpy() for k, v in g.items()})
            lrlog[names[i]].append(lr_a[i])
    np_ = lambda d: {k: v.detach().cpu().numpy() for k, v in d.items()}
    sds, tgt = net.state_dicts(), net.target_state_dicts()
    adam_path_close("critic", np_(sds["critic"]), np_(oracle.critic), glog["critic"], lrlog["critic"], g_floor=1e-5, zero_is_exact=True)
    adam_path_close("critic_target", np_(tgt["critic"]), np_(oracle.critic_target), glog["critic"], lrlog["critic"], g_floor=1e-5, zero_is_exact=True)
    for i, n in enumerate(names):
        adam_path_close(n, np_(sds[n]), np_(oracle.actors[i]), glog[n], lrlog[n], g_floor=2e-5, zero_is_exact=True)
        adam_path_close(n + "_target", np_(tgt[n]), np_(oracle.actor_targets[i]), glog[n], lrlog[n], g_floor=2e-5, zero_is_exact=True)


def test_td3_store_and_sample_surface():
    """store_transition layouts, pointer / learn_begin bookkeeping and an
    un-injected learn() (own indices, noise and order) run end to end."""
    import hhrl_vae_amd.algorithm as A
    cfg = gen.cav_cfg(o=5, g=3, hidden=(16, 12), batch=8, cap=20)
    w = A.WorkerTD3(cfg)
    assert w.memory.shape == (20, 2 * (5 + 3) + 1 + 1) and w.learn_begin == 20 * 0.4
    rs = np.random.RandomState(0)
    for i in range(25):                                    # wraps around the ring
        w.store_transition(rs.rand(5), rs.rand(1), rs.rand(3), rs.rand(3), [0.5], rs.rand(5))
    assert w.pointer == 25 and np.all(w.memory[:, 12] == 0.5)
    before = w.state_dicts()["critic"]["l1.weight"].clone()
    w.learn(); w.learn()
    assert w.total_it == 2 and not torch.equal(before, w.state_dicts()["critic"]["l1.weight"])
    t = A.HATD3Triple(gen.light_cfg(o=(5, 6, 7), T=4, s=8, a=(1, 2, 3), actor_h=(12, 10), critic_h=(14, 12), batch=8, cap=16))
    assert t.memory.shape[1] == 2 * 4 * (5 + 6 + 7) + 6 + 1 and (t.a1_dim, t.a2_dim, t.a3_dim, t.t1) == (1, 2, 3, 4)
    for i in range(10):
        t.store_transition(rs.rand(20), rs.rand(24), rs.rand(28), rs.rand(1), rs.rand(2), rs.rand(3), 1.0,
                           rs.rand(20), rs.rand(24), rs.rand(28))
    t.learn(None); t.learn(None)
    m = A.ManagerTD3(gen.light_cfg(o=(5, 5, 7), T=4, s=8, a=(1, 2, 3), veh_h=(12, 10), batch=8, cap=16))
    for i in range(10):
        m.store_transition(rs.rand(28), rs.rand(3), 0.25, rs.rand(28), None, None)
    assert m.pointer == 10 and m.memory["reward"][3, 0] == 0.25
    m.learn(None); m.learn(None)
    mcfg, wcfg = gen.opc_cfgs()
    mo, wk = A.ManagerTD3(dict(mcfg, memory_capacity=16)), A.WorkerTD3(wcfg)
    for i in range(10):
        mo.store_transition(rs.rand(28), rs.rand(8), 0.25, rs.rand(28), rs.rand(25, 8, 8), rs.rand(25, 8, 1))
    assert mo.memory["worker_obs"].shape == (16, 25, 8, 8) and mo.memory["worker_act"][9].any() and not mo.memory["worker_act"][10].any()
    with pytest.raises(ValueError):
        mo.learn(None)                                     # relabelling needs the worker's policy
    mo.learn(wk.policy); mo.learn(wk.policy)
    assert mo.total_it == 2


def test_td3_manager_off_policy_correction():
    """ManagerTD3.learn(worker.policy) with use_opc=True against the reference (tests/golden/make_opc.py):
    numpy's global generator seeded as in the recording, so `learn` draws the same indices and the same
    candidate goals; the worker policy runs on the GPU in one batched call.  The relabelled goals must
    be the reference's exactly: they are candidates picked by an argmax, and the recording seeds were
    chosen so that every decision wins by >= 1e-4 in score (`margin` in the fixture) where fp32
    differences between policy implementations move scores by ~1e-6.  The update that follows is
    held to the usual fp32 bar."""
    import hhrl_vae_amd.algorithm as A
    fx = load_case("td3_manager_opc")
    mcfg, wcfg = gen.opc_cfgs()
    net, worker = A.ManagerTD3(mcfg), A.WorkerTD3(wcfg)
    net.load_state_dicts({"actor": sub(fx, "learn/init/actor"), "critic": sub(fx, "learn/init/critic")})
    worker.load_state_dicts({"actor": sub(fx, "learn/init/worker_actor")})
    rows = fx["learn/rows"]
    fill_memory(net, "manager", rows)
    n = rows.shape[0]
    net.memory["worker_obs"][:n], net.memory["worker_act"][:n] = fx["learn/worker_obs"], fx["learn/worker_act"]
    seen = []
    real = net.off_policy_corrections
    net.off_policy_corrections = lambda *a, **kw: seen.append(real(*a, **kw)) or seen[-1]
    for call in range(int(fx["learn/n_calls"])):
        pre = "learn/call%d" % call
        assert fx[pre + "/margin"].min() >= 1e-4
        np.random.seed(int(fx[pre + "/seed"]))
        net.learn(worker.policy, noise=[fx[pre + "/eps0"]])
        assert np.array_equal(seen[-1], fx[pre + "/corrected"]), pre
        scaled_close(pre + "/mse", net.last_critic_loss.cpu().numpy(), fx[pre + "/mse"])
        close_dict(pre + "/grad/critic", net.critic.opt.g.state_dict(), sub(fx, pre + "/grad/critic"), False, atol_rel=5e-6)
        close_dict(pre + "/param/critic", net.critic.params.state_dict(), sub(fx, pre + "/param/critic"), False)
        if bool(fx[pre + "/policy"]):
            close_dict(pre + "/grad/actor", net.actors[0].opt.g.state_dict(), sub(fx, pre + "/grad/actor"), False, atol_rel=5e-6)
            close_dict(pre + "/param/actor", net.actors[0].params.state_dict(), sub(fx, pre + "/param/actor"), False)
            close_dict(pre + "/target/actor", net.actor_targets[0].params.state_dict(), sub(fx, pre + "/target/actor"), False)
            close_dict(pre + "/target/critic", net.critic_target.params.state_dict(), sub(fx, pre + "/target/critic"), False)
    assert len(seen) == 4
    close_dict("exp_avg/critic", net.critic.opt.m.state_dict(), sub(fx, "learn/final/exp_avg/critic"), False, atol_rel=5e-6)
    close_dict("exp_avg_sq/critic", net.critic.opt.v.state_dict(), sub(fx, "learn/final/exp_avg_sq/critic"), False, atol_rel=1e-5)


def test_td3_device_replay_matches_host_gather():
    """device_replay=True (replay rows mirrored in device memory, minibatch gathered there from the sampled
    row numbers) gives bit-identical updates to the host gather: bulk fill, single stores, ring wrap-around,
    a direct overwrite announced with mark_replay_dirty()."""
    import hhrl_vae_amd.algorithm as A
    for make in ((lambda **kw: A.WorkerTD3(gen.cav_cfg(o=5, g=3, hidden=(16, 12), batch=8, cap=24), seed=5, **kw)),
                          (lambda **kw: A.HATD3Triple(gen.light_cfg(o=(5, 6, 7), T=4, s=8, a=(1, 2, 3), actor_h=(12, 10),
                                                                    critic_h=(14, 12), batch=8, cap=24), seed=5, **kw))):
        host, dev = make(), make(device_replay=True)
        assert dev.device_replay and not host.device_replay
        W = host.fam.row_width
        rs = np.random.RandomState(3)
        rows = rs.uniform(-1, 1, size=(20, W))
        for net in (host, dev):
            net.memory[:20] = rows
            net.pointer = 20
        step = [0]

        def both_learn():
            step[0] += 1
            idx = np.random.RandomState(100 + step[0]).choice(min(host.pointer, 24), size=8)
            noise = [np.random.RandomState(200 + step[0] + k).standard_normal((8, a)).astype(np.float32)
                     for k, a in enumerate(host.fam.a_dims)]
            order = np.random.RandomState(300 + step[0]).permutation(len(host.fam.a_dims))
            for net in (host, dev):
                net._learn(idx, noise, order)
            a, b = host.state_dicts(), dev.state_dicts()
            for n in a:
                for k in a[n]:
                    assert torch.equal(a[n][k], b[n][k]), (step[0], n, k)
        for _ in range(4):
            both_learn()
        for i in range(9):                                   # single stores, wrapping around the 24-row ring
            row = rs.uniform(-1, 1, size=W)
            for net in (host, dev):
                net._store(row)
            both_learn()
        assert host.pointer == 29
        patch = rs.uniform(-1, 1, size=(3, W))
        for net in (host, dev):
            net.memory[10:13] = patch                        # a direct write into rows that are already mirrored
        dev.mark_replay_dirty()
        for _ in range(3):
            both_learn()


def test_td3_learn_rows_host_and_device_rows_agree():
    """learn_rows on a minibatch given as a numpy array, as a host tensor and as a device tensor (the block the update
    fetches then holds noise and optimizer scalars only): bit-identical learners, with noise from the caller and with
    noise drawn on the device, through the eager first updates and the replayed ones."""
    import hhrl_vae_amd.algorithm as A
    for with_noise in (True, False):
        mk = lambda: A.HATD3Triple(gen.light_cfg(o=(5, 6, 7), T=4, s=8, a=(1, 2, 3), actor_h=(12, 10), critic_h=(14, 12),
                                                 batch=8, cap=24), seed=9, device_replay=False)
        nets = [mk(), mk(), mk()]
        W, dims = nets[0].fam.row_width, nets[0].fam.a_dims
        for it in range(12):
            rs = np.random.RandomState(500 + it)
            rows = rs.uniform(-1, 1, size=(8, W)).astype(np.float32)
            noise = [rs.standard_normal((8, a)).astype(np.float32) for a in dims] if with_noise else None
            forms = (rows, torch.from_numpy(rows), torch.from_numpy(rows).cuda())
            for net, r in zip(nets, forms):
                net.learn_rows(r, noise, list(range(len(dims))))
        assert all(set(n.launch_forms().values()) == {"graph"} for n in nets)
        ref = nets[0].state_dicts()
        for other in nets[1:]:
            sd = other.state_dicts()
            for n in ref:
                for k in ref[n]:
                    assert torch.equal(ref[n][k], sd[n][k]), (with_noise, n, k)


def test_td3_smoothing_noise_drawn_on_the_device():
    """Without noise from the caller a single-device learner draws the target-policy smoothing noise (algorithm.py:790,
    1333: torch.randn_like on the device) in the update's first launch: consecutive runs of the standard-normal stream
    of hhrl_randn_fill for the learner's seed, the same values whether the launch chain runs eagerly (first updates) or
    is replayed from its graphs; two learners with one seed stay bit-identical, host-gathered and device-resident replay
    alike."""
    import ctypes as C
    import hhrl_vae_amd.algorithm as A
    from hhrl_vae_amd import _lib as L
    lib = L.load()
    for make in ((lambda: A.WorkerTD3(gen.cav_cfg(o=5, g=3, hidden=(16, 12), batch=8, cap=24), seed=11)),
                 (lambda: A.HATD3Triple(gen.light_cfg(o=(5, 6, 7), T=4, s=8, a=(1, 2, 3), actor_h=(12, 10), critic_h=(14, 12),
                                                      batch=8, cap=24), seed=11, device_replay=True))):
        a, b = make(), make()
        rows = np.random.RandomState(3).uniform(-1, 1, size=(20, a.fam.row_width))
        for net in (a, b):
            net.memory[:20] = rows
            net.pointer = 20
        n_act = len(a.fam.a_dims)
        for it in range(14):
            idx = np.random.RandomState(50 + it).choice(20, size=8)
            for net in (a, b):
                net._learn(idx, None, list(range(n_act)))       # one actor order: both chains reach their replayed form
            st = a._statics[8]
            n_noise = st.hyper_off[0] - st.noise_off[0]
            assert a._noise_drawn == (it + 1) * n_noise
            want = torch.empty(n_noise, device=a.device)
            L.check(lib.hhrl_randn_fill(want.data_ptr(), n_noise, C.c_ulonglong(a._noise_seed), C.c_ulonglong(it * n_noise),
                                        L.stream_ptr()))
            for k, (o, ad) in enumerate(zip(st.noise_off, a.fam.a_dims)):
                got = st.noise[k].reshape(-1)
                assert torch.equal(got, want[o - st.noise_off[0]: o - st.noise_off[0] + 8 * ad]), (it, k)
        assert set(a.launch_forms().values()) == {"graph"}
        sa, sb = a.state_dicts(), b.state_dicts()
        for n in sa:
            for k in sa[n]:
                assert torch.equal(sa[n][k], sb[n][k]), (n, k)
        flat = torch.cat([x.reshape(-1) for x in a._statics[8].noise])
        assert torch.isfinite(flat).all() and flat.abs().max() < 6.0


def test_td3_alternating_global_batch_keeps_each_chains_constant():
    """Two global batch sizes alternating at one shard size, on a family that takes the layer-by-layer path
    (the policy step's dL/dQ1 = -1/global batch sits in a buffer that captured chains read): the replayed
    graphs must use their own constant, as the always-eager learner does."""
    import hhrl_vae_amd.algorithm as A
    mk = lambda **kw: A.TD3Single(gen.light_cfg(o=(5, 5, 7), T=4, s=8, a=(1, 2, 3), actor_h=(12, 10), critic_h=(14, 12),
                                                batch=8, cap=48), "time", seed=5, **kw)
    graph, eager = mk(use_graph=True), mk(use_graph=False)
    assert graph.critic.mlp3 is None
    rows = gen.gen_rows(40, graph.fam.row_width, 3)
    for net in (graph, eager):
        net.memory[:40] = rows
        net.pointer = 40
    rs = np.random.RandomState(4)
    for it in range(16):
        idx = rs.randint(0, 40, size=8)
        noise = [rs.standard_normal((8, a)).astype(np.float32) for a in graph.fam.a_dims]
        gb = 8 if (it // 2) % 2 == 0 else 24                 # policy steps (odd `it`) alternate between the two
        for net in (graph, eager):
            net.learn(indices=idx, noise=noise, global_batch=gb)
    assert set(graph.launch_forms().values()) == {"graph"} and len(graph.launch_forms()) == 4
    a, b = graph.state_dicts(), eager.state_dicts()
    for n in a:
        for k in a[n]:
            assert torch.equal(a[n][k], b[n][k]), (n, k)


def test_td3_device_replay_default():
    """Without the argument the device mirror is used where a minibatch is wide enough to pay for it
    (batch x row width >= 32768 floats): HATD3Triple at its reference size, not WorkerTD3 at batch 64."""
    import hhrl_vae_amd.algorithm as A
    assert A.HATD3Triple(gen.light_cfg(cap=64)).device_replay
    assert not A.WorkerTD3(gen.cav_cfg(cap=64)).device_replay
    assert A.WorkerTD3(gen.cav_cfg(cap=64, batch=4096)).device_replay
    assert not A.HATD3Triple(gen.light_cfg(cap=64), device_replay=False).device_replay
    m = A.ManagerTD3(gen.light_cfg(cap=64, batch=4096))                 # dict replay: no device mirror to gather from,
    assert not hasattr(m, "_dev_mem") and not hasattr(m, "_sync_replay")   # the host gather is used whatever the flag says


def test_td3_checkpoint_roundtrip(tmp_path):
    """save/load: the reference's file suffixes and torch formats
    (algorithm.py:1368-1382); a torch Adam accepts the optimizer file."""
    import hhrl_vae_amd.algorithm as A
    cfg = gen.cav_cfg(o=5, g=3, hidden=(16, 12), batch=8, cap=48)
    w = A.WorkerTD3(cfg)
    rows = gen.gen_rows(40, w.fam.row_width, 1)
    w.memory[:40] = rows
    w.pointer = 40
    w.learn(); w.learn()
    prefix = os.path.join(str(tmp_path), "cav_agent_x_ep_9")
    w.save(prefix)
    for suf in ("_actor", "_actor_optimizer", "_critic", "_critic_optimizer"):
        assert os.path.exists(prefix + suf)
    sd = torch.load(prefix + "_critic", weights_only=True)
    assert list(sd) == ["l1.weight", "l1.bias", "l2.weight", "l2.bias", "l3.weight", "l3.bias",
                        "l4.weight", "l4.bias", "l5.weight", "l5.bias", "l6.weight", "l6.bias"]
    ps = [torch.nn.Parameter(v.clone()) for v in sd.values()]
    topt = torch.optim.Adam(ps, lr=1e-3)
    topt.load_state_dict(torch.load(prefix + "_critic_optimizer", weights_only=True))
    assert float(topt.state[ps[0]]["step"]) == 2.0
    w2 = A.WorkerTD3(gen.cav_cfg(o=5, g=3, hidden=(16, 12), batch=8, cap=48))
    w2.load(prefix)
    a, b = w.state_dicts(), w2.state_dicts()
    for n in a:
        for k in a[n]:
            assert torch.equal(a[n][k], b[n][k])
    tgt = w2.target_state_dicts()
    assert all(torch.equal(tgt["actor"][k], b["actor"][k]) for k in b["actor"])      # targets re-created as copies
    assert w2.critic.opt.step_count == 2 and w2.actors[0].opt.step_count == 1
