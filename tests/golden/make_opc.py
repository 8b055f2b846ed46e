"""Writes tests/golden/td3_manager_opc.npz: the manager's off-policy correction of stored goals
(`use_opc=True`, /root/reference/src/algorithm.py:1007-1107, 1124-1132), captured by importing the
reference on CPU.  Runs only in the build container.

    python tests/golden/make_opc.py

Two parts:
  fn/*    `ManagerTD3.off_policy_corrections` alone, with a closed-form numpy worker policy whose
          coefficients are in the fixture (so the check needs no network and no GPU), on inputs that
          include steps without a controlled vehicle (first observation feature == -1);
  learn/* four `ManagerTD3.learn(worker.policy)` calls with use_opc=True against a reference
          `WorkerTD3` (T=1) with deterministic weights: sampled indices, the corrected goals that
          entered the update, smoothing noise, losses, gradients, parameters, targets.  Seeds are
          chosen so that every relabelling decision wins by >= 1e-4 in score over any candidate of
          a different value (`margin`), so an fp32 policy on other hardware picks the same goals.
Randomness: np.random.seed(s) right before each call fixes `choice` and then the candidate goals,
lane by lane.  The smoothing noise is recorded from `torch.randn_like` itself rather than replayed
from the seed: the corrected goals reach torch as a column-major array (the reshape at
algorithm.py:1105-1106 returns a transposed view), `randn_like` keeps those strides, and the noise a
given (row, lane) receives is therefore not the one a contiguous `randn(B, 8)` would give it.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from make_golden import HERE, META, REF, _RecF, adam_np, gen, load_np, put, sd_np

CASE = "manager_opc"
LANES, WO, WA, SEQ = 8, 8, 1, 25      # the reference's replay buffer shapes (algorithm.py:926-927)


def closed_form_policy(w, c):
    """a = tanh(sum_j o_j w_j + g c): row-wise, so independent of how calls are batched."""
    def policy(o, g):
        o = np.asarray(o, dtype=np.float64)
        g = np.asarray(g, dtype=np.float64)
        return np.tanh((o * w).sum(axis=1, keepdims=True) + g * c)
    return policy


def worker_traces(n, seed):
    """Worker observations / actions recorded during one manager step, for n stored transitions."""
    obs = gen.gen_rows(n * SEQ * LANES, WO, seed).reshape(n, SEQ, LANES, WO)
    act = gen.gen_rows(n * SEQ * LANES, WA, seed + 1).reshape(n, SEQ, LANES, WA)
    r = np.random.RandomState(seed + 2)
    empty = r.rand(n, SEQ, LANES) < 0.15          # no controlled vehicle on that lane at that step
    obs[empty, 0] = -1
    act[empty] = -1
    return obs.astype(np.float32), act.astype(np.float32)


def decision_margins(policy, goals, wobs, wact):
    """For every (row, lane): score of the winning candidate minus the best score among candidates of
    a DIFFERENT value, drawing from numpy's global generator exactly as the relabelling does.  Only
    used to choose recording seeds whose decisions are not within fp32 noise of a tie."""
    B, S, L, wo = wobs.shape
    out = np.zeros((B, L))
    for lane in range(L):
        s = wobs[:, :, lane].astype(np.float64)
        feat = s[:, :, -1:]
        change = (wobs[:, -1, lane] - wobs[:, 0, lane])[:, None, -1:]
        draws = np.random.normal(loc=change, scale=.5, size=(B, 8, 1)).clip(-1, 1)
        c = np.concatenate([goals[:, lane][:, None, None], change, draws], 1)
        rows = wobs[:, :, lane].reshape(B * S, wo)
        score = np.zeros((B, c.shape[1]))
        for k in range(c.shape[1]):
            seen = (c[:, k] + wobs[:, 0, lane, -1:])[:, None] - wobs[:, :, lane, -1:]
            a = np.array(policy(rows, seen.reshape(B * S, 1)), dtype=np.float64)
            a[rows[:, 0] == -1] = -1
            score[:, k] = -0.5 * ((a - wact[:, :, lane].reshape(B * S, 1)).reshape(B, S) ** 2).sum(-1)
        for b in range(B):
            best = int(np.argmax(score[b]))
            rivals = [score[b, k] for k in range(c.shape[1]) if c[b, k, 0] != c[b, best, 0]]
            out[b, lane] = score[b, best] - max(rivals) if rivals else np.inf
    return out


MIN_MARGIN = 1e-4     # scores are O(1); fp32 differences between policy implementations move them by ~1e-6


def make_fn(out):
    B = 6
    w = gen.gen_rows(1, WO, gen.case_seed(CASE, "fn_w"))[0].astype(np.float64)
    c = 0.7
    obs, act = worker_traces(B, gen.case_seed(CASE, "fn_traces") % (2 ** 31))
    # lane 2: nothing controlled at all; lane 5: only the very first row is marked
    obs[:, :, 2, 0] = -1
    act[:, :, 2] = -1
    obs[:, :, 5, 0] = np.where(obs[:, :, 5, 0] == -1, 0.25, obs[:, :, 5, 0])
    obs[0, 0, 5, 0] = -1
    goals = gen.gen_rows(B, LANES, gen.case_seed(CASE, "fn_goals")).astype(np.float32)
    seed = gen.case_seed(CASE, "fn_call") % (2 ** 31)
    np.random.seed(seed)
    res = REF.ManagerTD3.off_policy_corrections(closed_form_policy(w, c), goals, obs, act)
    out["fn/w"], out["fn/c"], out["fn/seed"] = w, np.float64(c), np.int64(seed)
    out["fn/goals"], out["fn/worker_obs"], out["fn/worker_act"] = goals, obs, act
    out["fn/corrected"] = res
    print("fn: corrected != stored on", int((res != goals).sum()), "of", res.size)


def make_learn(out):
    cfg, wcfg = gen.opc_cfgs()
    n_rows, n_calls = 40, 4
    net = REF.ManagerTD3(cfg)
    worker = REF.WorkerTD3(wcfg)
    shapes = {}
    for a in ("actor", "critic"):
        mod = getattr(net, a)
        shapes[a] = [(k, tuple(v.shape)) for k, v in mod.state_dict().items()]
        st = gen.gen_state(shapes[a], gen.case_seed(CASE, "init_" + a))
        load_np(mod, st)
        load_np(getattr(net, a + "_target"), st)
        put(out, "learn/init/" + a, st)
    wst = gen.gen_state([(k, tuple(v.shape)) for k, v in worker.actor.state_dict().items()],
                        gen.case_seed(CASE, "init_worker_actor"))
    load_np(worker.actor, wst)
    put(out, "learn/init/worker_actor", wst)

    od, ad = net.obs_dim, net.a_dim
    rows = gen.gen_rows(n_rows, 2 * od + ad + 1, gen.case_seed(CASE, "rows"))
    wobs, wact = worker_traces(n_rows, gen.case_seed(CASE, "traces") % (2 ** 31))
    buf = net.memory.buffer
    buf["obs"][:n_rows] = rows[:, :od]
    buf["action"][:n_rows] = rows[:, od: od + ad]
    buf["reward"][:n_rows] = rows[:, od + ad: od + ad + 1]
    buf["next_obs"][:n_rows] = rows[:, od + ad + 1:]
    buf["worker_obs"][:n_rows] = wobs
    buf["worker_act"][:n_rows] = wact
    net.memory.pointer = n_rows
    out["learn/rows"], out["learn/worker_obs"], out["learn/worker_act"] = rows, wobs, wact
    out["learn/n_rows"], out["learn/n_calls"] = np.int64(n_rows), np.int64(n_calls)

    rec = _RecF(torch.nn.functional)
    REF.F = rec
    snaps, corrected = {}, []

    def wrap(opt, label, mod):
        orig = opt.step

        def step(*a, **kw):
            snaps[label] = {k: p.grad.detach().numpy().copy() for k, p in mod.named_parameters()}
            return orig(*a, **kw)
        opt.step = step
    wrap(net.actor_optimizer, "actor", net.actor)
    wrap(net.critic_optimizer, "critic", net.critic)
    real_opc = REF.ManagerTD3.off_policy_corrections

    def spy(*a, **kw):
        r = real_opc(*a, **kw)
        corrected.append(np.array(r))
        return r
    net.off_policy_corrections = spy
    drawn = []
    real_randn_like = torch.randn_like

    def randn_like_spy(x, *a, **kw):
        r = real_randn_like(x, *a, **kw)
        drawn.append(r.numpy().copy())
        return r
    torch.randn_like = randn_like_spy

    B = cfg["batch_size"]
    for call in range(n_calls):
        for attempt in range(64):
            seed = gen.case_seed(CASE, "call%d_%d" % (call, attempt)) % (2 ** 31)
            np.random.seed(seed)
            idx = np.random.choice(n_rows, size=B)
            margin = decision_margins(worker.policy, buf["action"][idx], wobs[idx], wact[idx])
            if margin.min() >= MIN_MARGIN:
                break
        else:
            raise RuntimeError("no seed with clear decisions")
        np.random.seed(seed)
        idx = np.random.choice(n_rows, size=B)
        policy = (net.total_it + 1) % net.policy_freq == 0
        np.random.seed(seed)
        torch.manual_seed(seed)
        rec.mse.clear()
        snaps.clear()
        corrected.clear()
        drawn.clear()
        net.learn(worker.policy)
        eps = drawn[0]
        assert eps.shape == (B, ad) and len(drawn) == 1
        pre = "learn/call%d" % call
        out[pre + "/seed"] = np.int64(seed)
        out[pre + "/indices"] = idx
        out[pre + "/policy"] = np.bool_(policy)
        out[pre + "/eps0"] = eps
        out[pre + "/corrected"] = corrected[0]
        out[pre + "/stored"] = buf["action"][idx]
        out[pre + "/margin"] = margin
        out[pre + "/mse"] = np.array(rec.mse[:2])
        put(out, pre + "/grad/critic", snaps["critic"])
        put(out, pre + "/param/critic", sd_np(net.critic.state_dict()))
        if policy:
            put(out, pre + "/grad/actor", snaps["actor"])
            put(out, pre + "/param/actor", sd_np(net.actor.state_dict()))
            put(out, pre + "/target/actor", sd_np(net.actor_target.state_dict()))
            put(out, pre + "/target/critic", sd_np(net.critic_target.state_dict()))
        print("call", call, "attempt", attempt, "min margin %.2e" % margin.min(), "relabelled", int((corrected[0] != buf["action"][idx]).sum()), "of", corrected[0].size)
    for a in ("actor", "critic"):
        m, v = adam_np(getattr(net, a + "_optimizer"), [k for k, _ in shapes[a]])
        put(out, "learn/final/exp_avg/" + a, m)
        put(out, "learn/final/exp_avg_sq/" + a, v)
    REF.F = torch.nn.functional
    torch.randn_like = real_randn_like


if __name__ == "__main__":
    out = {"meta": META}
    make_fn(out)
    make_learn(out)
    np.savez_compressed(os.path.join(HERE, "td3_" + CASE + ".npz"), **out)
    print("wrote td3_" + CASE)
