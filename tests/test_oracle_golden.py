"""Pins the CPU oracle to golden vectors recorded from the reference
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

import gen
import hhrl_oracle as O
from golden_util import assert_close, assert_dict_close, load_case, sub, td3_inputs

# Same ATen operators on both sides, but not necessarily the same CPU: the BLAS picks its blocking (the order of
# the fp32 sums) by micro-architecture, so the fixtures recorded on one host are met to summation noise on another.
# Still far tighter than the product bar (rtol 1e-4 + 1e-5 of the tensor's largest entry).
TOL = dict(rtol=2e-5, atol=1e-7, max_frac=5e-6)


def PTOL(lr, steps):
    """Parameters after `steps` Adam steps: a step moves a weight by about lr * g/|g|, which turns the summation
    noise of a near-zero gradient element into a difference of a small share of lr."""
    return dict(rtol=2e-5, atol=1e-7 + 1e-3 * lr * steps, max_frac=5e-6)


def _t(d):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in d.items()}


@pytest.mark.parametrize("g", [2, 4, 8, 16])
def test_vae_steps(g):
    fx = load_case("vae_g%d" % g)
    vae = O.VAE(_t(sub(fx, "init")), g)
    assert [n for n, _ in O.vae_param_shapes(g)] == list(sub(fx, "init").keys())
    for step in range(3):
        loss = vae.learn(fx["step%d/x" % step], torch.from_numpy(fx["step%d/eps" % step]))
        assert_close("loss%d" % step, loss, fx["step%d/loss" % step], **TOL)
        if step == 0 and g == 4:
            assert_dict_close("grad0", {k: v.numpy() for k, v in vae.grads.items()}, sub(fx, "step0/grad"), **TOL)
            assert_dict_close("param0", {k: v.detach().numpy() for k, v in vae.p.items()}, sub(fx, "step0/param"), **PTOL(1e-3, 1))
    assert_dict_close("final", {k: v.detach().numpy() for k, v in vae.p.items()}, sub(fx, "final/param"), **PTOL(1e-3, 3))
    if g == 4:
        m = {n: t.numpy() for n, t in zip(vae.names, vae.opt.exp_avg)}
        v = {n: t.numpy() for n, t in zip(vae.names, vae.opt.exp_avg_sq)}
        assert_dict_close("exp_avg", m, sub(fx, "final/exp_avg"), **TOL)
        assert_dict_close("exp_avg_sq", v, sub(fx, "final/exp_avg_sq"), rtol=2e-5, atol=1e-12)


@pytest.mark.parametrize("tag", ["edge_n1", "edge_b1", "edge_n59"])
def test_vae_ragged(tag):
    fx = load_case("vae_g4")
    vae = O.VAE(_t(sub(fx, "final/param")), 4)
    loss, grads = vae.loss_and_grads(torch.from_numpy(fx[tag + "/x"]), torch.from_numpy(fx[tag + "/eps"]))
    assert_close("loss", float(loss), fx[tag + "/loss"], **TOL)
    assert_dict_close("grad", {k: v.numpy() for k, v in grads.items()}, sub(fx, tag + "/grad"), **TOL)


def test_vae_encode_decode():
    fx = load_case("vae_g4")
    vae = O.VAE(_t(sub(fx, "final/param")), 4)
    assert_close("encode", vae.encode(fx["encode/x"]), fx["encode/out"], **TOL)
    assert np.array_equal(vae.encode([]), fx["encode/empty"]) and fx["encode/empty"].shape == (8,)
    out = vae.decode(fx["decode/goal"], fx["decode/loc"], torch.from_numpy(fx["decode/eps"]))
    assert_close("decode", out, fx["decode/out"], **TOL)


def test_vae_batch_loop():
    fx = load_case("vae_loop_g2")
    dataset = {k: v.tolist() for k, v in sub(fx, "bucket").items()}
    chunks = O.vae_chunk_dataset(dataset, int(fx["batch_size"]))
    order = fx["order"]
    assert [len(chunks[i]) for i in order] == fx["chunk_sizes"].tolist()
    vae = O.VAE(_t(sub(fx, "init")), 2)
    losses = [vae.learn(chunks[ci], torch.from_numpy(fx["eps%d" % s])) for s, ci in enumerate(order)]
    assert_close("losses", losses, fx["losses"], **TOL)
    assert_dict_close("final", {k: v.detach().numpy() for k, v in vae.p.items()}, sub(fx, "final/param"),
                      **PTOL(1e-3, len(order)))


def test_lstm_matches_nn_lstm():
    """The hand-written BiLSTM restatement against torch.nn.LSTM."""
    torch.manual_seed(0)
    o, T, h, B = 7, 5, 12, 9
    ref = torch.nn.LSTM(o, h, 1, batch_first=True, bidirectional=True)
    fc = torch.nn.Linear(2 * h, h)
    p = {"lstm.lstm_layer." + k: v.detach() for k, v in ref.state_dict().items()}
    p.update({"lstm.fc.weight": fc.weight.detach(), "lstm.fc.bias": fc.bias.detach()})
    x = torch.randn(B, T * o)
    want = fc(ref(x.view(-1, T, o))[0][:, -1, :])
    got = O.lstm_forward(p, "lstm.", x, T, o)
    assert_close("lstm", got.detach().numpy(), want.detach().numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("case", list(gen.TD3_CASES))
def test_td3_case(case):
    kind, task, mkcfg, n_rows, n_calls, mode = gen.TD3_CASES[case]
    fx = load_case("td3_" + case)
    fam = O.family_from_cfg(kind, mkcfg(), task)
    names = fam.actor_names()
    shapes = {n: O.actor_param_shapes(fam, k) for k, n in enumerate(names)}
    shapes["critic"] = O.critic_param_shapes(fam)
    init, rows = td3_inputs(case, fx, shapes)
    assert rows.shape[1] == fam.row_width
    net = O.TD3(fam, [_t(init[n]) for n in names], _t(init["critic"]))
    dg = mode == "digest"
    npd = lambda d: {k: v.detach().numpy() for k, v in d.items()}
    for call in range(n_calls):
        pre = "call%d" % call
        b = O.split_rows(fam, rows[fx[pre + "/indices"]])
        noise = [torch.from_numpy(fx[pre + "/eps%d" % j]) for j in range(len(names))]
        out = net.learn(b, noise, fx[pre + "/order"])
        assert_close(pre + "/critic_loss", out["critic_loss"], fx[pre + "/mse"].sum(), **TOL)
        assert_dict_close(pre + "/grad/critic", npd(out["critic_grads"]), sub(fx, pre + "/grad/critic"), digest=dg, **TOL)
        assert_dict_close(pre + "/param/critic", npd(net.critic), sub(fx, pre + "/param/critic"), digest=dg, **TOL)
        assert bool(fx[pre + "/policy"]) == ("actor_grads" in out)
        if "actor_grads" in out:
            for k, n in enumerate(names):
                assert_dict_close(pre + "/grad/" + n, npd(out["actor_grads"][k]), sub(fx, pre + "/grad/" + n), digest=dg, **TOL)
                assert_dict_close(pre + "/param/" + n, npd(net.actors[k]), sub(fx, pre + "/param/" + n), digest=dg, **TOL)
                assert_dict_close(pre + "/target/" + n, npd(net.actor_targets[k]), sub(fx, pre + "/target/" + n), digest=dg, **TOL)
            assert_dict_close(pre + "/target/critic", npd(net.critic_target), sub(fx, pre + "/target/critic"), digest=dg, **TOL)
        assert_close(pre + "/lr_critic", net.critic_opt.lr, fx[pre + "/lr_critic"], rtol=1e-12, atol=0)
        assert_close(pre + "/lr_actor", net.actor_opts[0].lr, fx[pre + "/lr_actor"], rtol=1e-12, atol=0)
    for k, n in enumerate(names + ["critic"]):
        opt = net.critic_opt if n == "critic" else net.actor_opts[k]
        pn = net.critic_names if n == "critic" else net.actor_names[k]
        assert_dict_close("exp_avg/" + n, {a: t.numpy() for a, t in zip(pn, opt.exp_avg)}, sub(fx, "final/exp_avg/" + n), digest=dg, **TOL)
        assert_dict_close("exp_avg_sq/" + n, {a: t.numpy() for a, t in zip(pn, opt.exp_avg_sq)}, sub(fx, "final/exp_avg_sq/" + n),
                          digest=dg, rtol=2e-5, atol=1e-12)
    if not dg:
        for k, n in enumerate(names):
            o = fx["choose/%s/o" % n]
            g = fx.get("choose/%s/g" % n)
            assert_close("choose/" + n, net.choose(k, o, g), fx["choose/%s/out" % n], **TOL)
        if "policy/out" in fx:
            with torch.no_grad():
                got = O.actor_forward(net.actors[0], fam, 0, torch.from_numpy(fx["policy/o"].astype(np.float32)),
                                      torch.from_numpy(fx["policy/g"].astype(np.float32))).numpy()
            assert_close("policy", got, fx["policy/out"], **TOL)


def test_td3_manager_off_policy_correction():
    """Manager updates with use_opc=True (tests/golden/make_opc.py): the oracle's relabelling with the
    oracle's worker policy picks the reference's goals exactly, and the update that follows matches."""
    fx = load_case("td3_manager_opc")
    mcfg, wcfg = gen.opc_cfgs()
    fam, wfam = O.family_from_cfg("manager", mcfg), O.family_from_cfg("worker", wcfg)
    worker = O.TD3(wfam, [_t(sub(fx, "learn/init/worker_actor"))],
                   {k: torch.zeros(s) for k, s in O.critic_param_shapes(wfam)})
    net = O.TD3(fam, [_t(sub(fx, "learn/init/actor"))], _t(sub(fx, "learn/init/critic")))
    rows, wobs, wact = fx["learn/rows"], fx["learn/worker_obs"], fx["learn/worker_act"]
    od, ad = fam.streams[0].width, fam.a_dims[0]
    npd = lambda d: {k: v.detach().numpy() for k, v in d.items()}
    for call in range(int(fx["learn/n_calls"])):
        pre = "learn/call%d" % call
        np.random.seed(int(fx[pre + "/seed"]))
        idx = np.random.choice(rows.shape[0], size=mcfg["batch_size"])
        assert np.array_equal(idx, fx[pre + "/indices"])
        batch = rows[idx].astype(np.float32)
        goals = O.off_policy_corrections(worker.policy, batch[:, od: od + ad], wobs[idx], wact[idx])
        assert np.array_equal(goals, fx[pre + "/corrected"])
        batch[:, od: od + ad] = goals
        out = net.learn(O.split_rows(fam, batch), [torch.from_numpy(fx[pre + "/eps0"])])
        assert_close(pre + "/critic_loss", out["critic_loss"], fx[pre + "/mse"].sum(), **TOL)
        assert_dict_close(pre + "/grad/critic", npd(out["critic_grads"]), sub(fx, pre + "/grad/critic"), **TOL)
        assert_dict_close(pre + "/param/critic", npd(net.critic), sub(fx, pre + "/param/critic"), **TOL)
        if bool(fx[pre + "/policy"]):
            assert_dict_close(pre + "/grad/actor", npd(out["actor_grads"][0]), sub(fx, pre + "/grad/actor"), **TOL)
            assert_dict_close(pre + "/param/actor", npd(net.actors[0]), sub(fx, pre + "/param/actor"), **TOL)
            assert_dict_close(pre + "/target/critic", npd(net.critic_target), sub(fx, pre + "/target/critic"), **TOL)
