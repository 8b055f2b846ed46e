"""Writes the golden vectors under tests/golden/ by importing the reference
(`/root/reference/src/algorithm.py`) on CPU.  Runs only in the build
container; the fixtures it writes are what travels.

    python tests/golden/make_golden.py            # all cases
    python tests/golden/make_golden.py vae td3    # a subset

How randomness inside the reference is captured without editing it:
  * torch.manual_seed(s) right before a call, then the same seed replayed
    with torch.randn(shape) outside, reproduces every `randn_like` draw;
  * np.random.seed(s) likewise for `np.random.choice` / `permutation`;
  * optimizer.step of each torch optimizer is wrapped to snapshot the
    gradients it is about to apply (gradients of multi-actor variants are
    otherwise polluted by later backward passes);
  * F.mse_loss as seen by the reference module is wrapped to record the two
    critic losses.
Fixture metadata records the torch / numpy versions used.
"""
from __future__ import annotations

import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference/src")

import gen  # noqa: E402
import algorithm as REF  # noqa: E402  (the reference)

torch.set_num_threads(1)
META = np.array([torch.__version__, np.__version__, "Mamo666/HHRL-VAE@2024_10_08"])


def sd_np(sd):
    return {k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


def load_np(module, state):
    module.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})


def adam_np(opt, names):
    """exp_avg / exp_avg_sq by parameter name (state index order == names order)."""
    st = opt.state_dict()["state"]
    m = {n: st[i]["exp_avg"].numpy().copy() for i, n in enumerate(names) if i in st}
    v = {n: st[i]["exp_avg_sq"].numpy().copy() for i, n in enumerate(names) if i in st}
    return m, v


def put(out, prefix, d):
    for k, v in d.items():
        out[prefix + "/" + k] = np.asarray(v)


# ------------------------------------------------------------------ VAE
def vae_shapes(g):
    e = REF.Encoder(2, g)
    return [(k, tuple(v.shape)) for k, v in e.state_dict().items()]


def make_vae():
    for g in (2, 4, 8, 16):
        case = "vae_g%d" % g
        enc = REF.Encoder(2, g)
        shapes = [(k, tuple(v.shape)) for k, v in enc.state_dict().items()]
        names = [k for k, _ in shapes]
        init = gen.gen_state(shapes, gen.case_seed(case, "init"))
        load_np(enc, init)
        out = {"meta": META, "goal_dim": np.int64(g)}
        put(out, "init", init)
        B, N = 8, 5
        for step in range(3):
            x = gen.gen_lanes(B, N, gen.case_seed(case, "x%d" % step))
            seed = gen.case_seed(case, "eps%d" % step)
            torch.manual_seed(seed)
            eps = torch.randn(B, g).numpy().copy()
            torch.manual_seed(seed)
            loss = enc.learn(x.tolist())
            out["step%d/x" % step] = x
            out["step%d/eps" % step] = eps
            out["step%d/loss" % step] = np.float64(loss)
            if step == 0 and g == 4:
                put(out, "step0/grad", {n: p.grad.numpy().copy() for n, p in zip(names, enc.parameters())})
                put(out, "step0/param", sd_np(enc.state_dict()))
        put(out, "final/param", sd_np(enc.state_dict()))
        if g == 4:
            m, v = adam_np(enc.opt, names)
            put(out, "final/exp_avg", m)
            put(out, "final/exp_avg_sq", v)
            # ragged edge cases on the trained weights: one car per lane, one lane
            for tag, (b, n) in {"edge_n1": (3, 1), "edge_b1": (1, 12), "edge_n59": (2, 59)}.items():
                e2 = REF.Encoder(2, g)
                load_np(e2, sd_np(enc.state_dict()))
                x = gen.gen_lanes(b, n, gen.case_seed(case, tag))
                seed = gen.case_seed(case, tag + "eps")
                torch.manual_seed(seed)
                eps = torch.randn(b, g).numpy().copy()
                torch.manual_seed(seed)
                loss = e2.learn(x.tolist())
                out[tag + "/x"], out[tag + "/eps"], out[tag + "/loss"] = x, eps, np.float64(loss)
                put(out, tag + "/grad", {n_: p.grad.numpy().copy() for n_, p in zip(names, e2.parameters())})
            # inference surface
            lane = gen.gen_lanes(1, 6, gen.case_seed(case, "lane"))[0]
            out["encode/x"] = lane
            out["encode/out"] = enc.encode(lane.tolist())
            out["encode/empty"] = enc.encode([])
            goal = out["encode/out"].astype(np.float32)
            loc = np.array([0.37], dtype=np.float32)
            seed = gen.case_seed(case, "decode")
            torch.manual_seed(seed)
            deps = torch.randn(g).numpy().copy()
            torch.manual_seed(seed)
            out["decode/goal"], out["decode/loc"], out["decode/eps"] = goal, loc, deps
            out["decode/out"] = enc.decode(goal, loc)
        np.savez_compressed(os.path.join(HERE, case + ".npz"), **out)
        print("wrote", case)


def make_vae_loop():
    """The batch loop of VAE_train.py:52-69 on a small synthetic state log:
    bucket by car count, chunk to <= 64 lanes, shuffle, one Adam step per
    chunk on a fresh model.  (The script itself cannot be imported: it needs
    traci/tensorboard and a data file that is not in the tree.)"""
    case = "vae_loop_g2"
    g = 2
    counts = {"1": 70, "2": 5, "3": 64, "4": 130, "5": 0}
    dataset = {}
    for k, c in counts.items():
        n = int(k)
        dataset[k] = gen.gen_lanes(c, n, gen.case_seed(case, "bucket" + k)).tolist() if c else []
    batch_size = 64
    chunked = {k: [v[i: i + batch_size] for i in range(0, len(v), batch_size)] for k, v in dataset.items() if len(v) > 0}
    flat = [item for sub in chunked.values() for item in sub]
    order = list(range(len(flat)))
    random.Random(3407).shuffle(order)
    enc = REF.Encoder(2, g)
    shapes = [(k, tuple(v.shape)) for k, v in enc.state_dict().items()]
    init = gen.gen_state(shapes, gen.case_seed(case, "init"))
    load_np(enc, init)
    out = {"meta": META, "goal_dim": np.int64(g), "order": np.array(order), "batch_size": np.int64(batch_size)}
    for k, c in counts.items():
        out["bucket/" + k] = np.asarray(dataset[k], dtype=np.float32).reshape(c, int(k), 2)
    put(out, "init", init)
    losses, sizes = [], []
    for step, ci in enumerate(order):
        data = flat[ci]
        seed = gen.case_seed(case, "eps%d" % step)
        torch.manual_seed(seed)
        eps = torch.randn(len(data), g).numpy().copy()
        torch.manual_seed(seed)
        losses.append(enc.learn(data))
        sizes.append(len(data))
        out["eps%d" % step] = eps
    out["losses"] = np.array(losses)
    out["chunk_sizes"] = np.array(sizes)
    put(out, "final/param", sd_np(enc.state_dict()))
    np.savez_compressed(os.path.join(HERE, case + ".npz"), **out)
    print("wrote", case, "chunks", sizes)


# ------------------------------------------------------------------ TD3
class _RecF:
    """Stands in for `torch.nn.functional` inside the reference module and
    records every mse_loss value."""

    def __init__(self, real):
        self._real = real
        self.mse = []

    def __getattr__(self, k):
        return getattr(self._real, k)

    def mse_loss(self, *a, **kw):
        v = self._real.mse_loss(*a, **kw)
        self.mse.append(float(v))
        return v


def build_ref(kind, task, cfg):
    if kind == "single":
        return REF.TD3Single(cfg, task)
    if kind == "double":
        return REF.HATD3Double(cfg, task)
    if kind == "tp":
        return REF.HATD3(cfg)
    if kind == "triple":
        return REF.HATD3Triple(cfg)
    if kind == "manager":
        return REF.ManagerTD3(cfg)
    if kind == "worker":
        return REF.WorkerTD3(cfg)
    raise ValueError(kind)


def actor_attrs(net):
    return [a for a in ("actor", "actor1", "actor2", "actor3") if hasattr(net, a)]


def make_td3(only=None):
    for case, (kind, task, mkcfg, n_rows, n_calls, mode) in gen.TD3_CASES.items():
        if only and case not in only:
            continue
        cfg = mkcfg()
        net = build_ref(kind, task, cfg)
        acts = actor_attrs(net)
        out = {"meta": META, "mode": np.array(mode), "n_calls": np.int64(n_calls), "n_rows": np.int64(n_rows)}
        store = (lambda pre, d: put(out, pre, d)) if mode == "full" else \
                (lambda pre, d: put(out, pre, {k: gen.digest(v) for k, v in d.items()}))
        # deterministic weights
        shapes = {}
        for a in acts + ["critic"]:
            mod = getattr(net, a)
            shp = [(k, tuple(v.shape)) for k, v in mod.state_dict().items()]
            shapes[a] = shp
            st = gen.gen_state(shp, gen.case_seed(case, "init_" + a))
            load_np(mod, st)
            load_np(getattr(net, a + "_target"), st)
            if mode == "full":
                put(out, "init/" + a, st)
        # replay contents
        B = cfg["batch_size"]
        if kind == "manager":
            width = 2 * net.obs_dim + net.a_dim + 1
        else:
            width = net.memory.shape[1]
        rows = gen.gen_rows(n_rows, width, gen.case_seed(case, "rows"))
        if kind == "manager":
            od, ad = net.obs_dim, net.a_dim
            buf = net.memory.buffer
            buf["obs"][:n_rows] = rows[:, :od]
            buf["action"][:n_rows] = rows[:, od: od + ad]
            buf["reward"][:n_rows] = rows[:, od + ad: od + ad + 1]
            buf["next_obs"][:n_rows] = rows[:, od + ad + 1:]
            net.memory.pointer = n_rows
        else:
            net.memory[:n_rows] = rows
            net.pointer = n_rows
        out["row_width"] = np.int64(width)
        if mode == "full":
            out["rows"] = rows
        # hooks
        rec = _RecF(torch.nn.functional)
        REF.F = rec
        snaps = {}

        def wrap(opt, label, mod):
            orig = opt.step

            def step(*a, **kw):
                snaps[label] = {k: p.grad.detach().numpy().copy() for k, p in mod.named_parameters()}
                return orig(*a, **kw)
            opt.step = step
        for a in acts:
            wrap(getattr(net, a + "_optimizer"), a, getattr(net, a))
        wrap(net.critic_optimizer, "critic", net.critic)

        a_dims = [getattr(net, a).l3.out_features for a in acts]
        for call in range(n_calls):
            seed = gen.case_seed(case, "call%d" % call)
            np.random.seed(seed)
            idx = np.random.choice(n_rows, size=B)
            policy = (net.total_it + 1) % net.policy_freq == 0
            order = np.random.permutation(len(acts)) if (policy and len(acts) > 1) else np.arange(len(acts))
            torch.manual_seed(seed)
            eps = [torch.randn(B, d).numpy().copy() for d in a_dims]
            # the call itself
            np.random.seed(seed)
            torch.manual_seed(seed)
            rec.mse.clear()
            snaps.clear()
            if kind in ("single", "worker"):
                net.learn()
            elif kind in ("double", "tp"):
                net.learn()
            else:
                net.learn(None)
            pre = "call%d" % call
            out[pre + "/indices"] = idx
            out[pre + "/order"] = np.asarray(order)
            out[pre + "/policy"] = np.bool_(policy)
            for j, e in enumerate(eps):
                out[pre + "/eps%d" % j] = e
            out[pre + "/mse"] = np.array(rec.mse[:2])
            store(pre + "/grad/critic", snaps["critic"])
            store(pre + "/param/critic", sd_np(net.critic.state_dict()))
            if policy:
                for a in acts:
                    store(pre + "/grad/" + a, snaps[a])
                    store(pre + "/param/" + a, sd_np(getattr(net, a).state_dict()))
                    store(pre + "/target/" + a, sd_np(getattr(net, a + "_target").state_dict()))
                store(pre + "/target/critic", sd_np(net.critic_target.state_dict()))
            out[pre + "/lr_critic"] = np.float64(net.critic_optimizer.param_groups[0]["lr"])
            out[pre + "/lr_actor"] = np.float64(getattr(net, acts[0] + "_optimizer").param_groups[0]["lr"])
        # final optimizer moments
        for a in acts + ["critic"]:
            names = [k for k, _ in shapes[a]]
            m, v = adam_np(getattr(net, a + "_optimizer"), names)
            store("final/exp_avg/" + a, m)
            store("final/exp_avg_sq/" + a, v)
        # batch-1 inference surface on the trained actors
        if mode == "full":
            for j, a in enumerate(acts):
                mod = getattr(net, a)
                if kind == "worker":
                    o = gen.gen_rows(1, net.obs_dim, gen.case_seed(case, "choose_o"))[0]
                    g_ = gen.gen_rows(1, net.g_dim, gen.case_seed(case, "choose_g"))[0]
                    out["choose/%s/o" % a], out["choose/%s/g" % a] = o, g_
                    out["choose/%s/out" % a] = net.choose_action(o, g_)
                    ob = gen.gen_rows(5, net.obs_dim, gen.case_seed(case, "policy_o"))
                    gb = gen.gen_rows(5, net.g_dim, gen.case_seed(case, "policy_g"))
                    if net.t == 1:
                        out["policy/o"], out["policy/g"] = ob, gb
                        out["policy/out"] = net.policy(ob, gb)
                else:
                    width_j = mod.l1.in_features if not hasattr(mod, "lstm") else mod.lstm.obs_dim * mod.lstm.T
                    o = gen.gen_rows(1, width_j, gen.case_seed(case, "choose_o%d" % j))[0]
                    fn = {"single": ["choose_action"], "manager": ["choose_action"],
                          "double": ["choose_light_action", "choose_goal"],
                          "tp": ["choose_time_action", "choose_phase_action"],
                          "triple": ["choose_time_action", "choose_phase_action", "choose_goal"]}[kind][j]
                    out["choose/%s/o" % a] = o
                    out["choose/%s/out" % a] = getattr(net, fn)(o)
        REF.F = torch.nn.functional
        np.savez_compressed(os.path.join(HERE, "td3_" + case + ".npz"), **out)
        print("wrote td3_" + case, "mode", mode)


if __name__ == "__main__":
    what = set(sys.argv[1:]) or {"vae", "loop", "td3"}
    if "vae" in what:
        make_vae()
    if "loop" in what:
        make_vae_loop()
    if "td3" in what:
        make_td3()
    extra = [w for w in what if w not in ("vae", "loop", "td3")]
    if extra:
        make_td3(only=set(extra))
