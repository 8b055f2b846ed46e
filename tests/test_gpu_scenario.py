"""Four-intersection scenario (tools/four_intersections.py) on the GPU classes against the same stream
driven through the CPU oracle: the reference's per-second call cadence (train.py:183-204, agent.py:309-494,
736-867) with full-size networks — Encoder(2, 2), HATD3Triple (13/13/25 x T=10, actions 1/4/4), WorkerTD3
(11 + goal 6) — replay filling, exploration, `learn` calls of both levels interleaved with `encode`,
`decode` and `choose_*`.  Every value that flows between calls is compared, and the parameters at the end.
Bar: fp32, rtol 1e-4 with the absolute floors of the other tests; parameters after their Adam steps are held
to golden_util.adam_path_close, whose allowance per element is built from the oracle's own gradient at every
update (large where that gradient is lost in rounding noise, nil where it is well conditioned) instead of a
term that grows with the number of steps for every element alike."""
import os
import sys

import numpy as np
import pytest
import torch

import hhrl_oracle as O
from golden_util import adam_path_close, assert_close
from test_gpu_kernels import scaled_close

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import four_intersections as S  # noqa: E402

pytestmark = pytest.mark.gpu


class OracleNet:
    """The reference's replay and call surface (algorithm.py:252, 273-281, 709, 750-766, 1276, 1299-1314)
    around the oracle's functional TD3."""

    def __init__(self, kind, cfg, states):
        self.fam = O.family_from_cfg(kind, cfg)
        names = self.fam.actor_names()
        self.td3 = O.TD3(self.fam, [states[n] for n in names], states["critic"])
        self.batch_size = self.fam.batch_size
        self.memory_capacity = cfg["memory_capacity"]
        self.learn_begin = self.memory_capacity * cfg["learn_start_ratio"]
        self.memory = np.zeros((self.memory_capacity, self.fam.row_width))
        self.pointer = 0
        # the gradient and the learning rate of every optimizer step, per network: the bar on the final parameters
        self.grad_log = {n: [] for n in names + ["critic"]}
        self.lr_log = {n: [] for n in names + ["critic"]}

    def store_transition(self, *parts):
        self.memory[self.pointer % self.memory_capacity, :] = np.hstack(parts)
        self.pointer += 1

    def choose_time_action(self, o):
        return self.td3.choose(0, o)

    def choose_phase_action(self, o):
        return self.td3.choose(1, o)

    def choose_goal(self, o):
        return self.td3.choose(2, o)

    def choose_goal_batch(self, rows):
        return np.stack([self.td3.choose(2, o) for o in rows])

    def choose_action(self, o, g):
        return self.td3.choose(0, o, g)

    def policy(self, o, g):
        return self.td3.policy(o, g)

    def learn(self, worker_policy=None, indices=None, noise=None, order=None):
        b = O.split_rows(self.fam, self.memory[indices, :])
        names = self.fam.actor_names()
        lr_c, lr_a = self.td3.critic_opt.lr, [o.lr for o in self.td3.actor_opts]
        out = self.td3.learn(b, [torch.as_tensor(n) for n in noise], order)
        self.grad_log["critic"].append({k: v.numpy() for k, v in out["critic_grads"].items()})
        self.lr_log["critic"].append(lr_c)
        for i, g in out.get("actor_grads", {}).items():
            self.grad_log[names[i]].append({k: v.numpy() for k, v in g.items()})
            self.lr_log[names[i]].append(lr_a[i])


class OracleEncoder:
    def __init__(self, state, g):
        self.goal_dim = g
        self.vae = O.VAE(state, g)

    def encode(self, x):
        return self.vae.encode(x)

    def decode(self, goal, loc, eps=None):
        return self.vae.decode(np.asarray(goal).reshape(1, -1), np.asarray(loc).reshape(1, 1),
                               torch.as_tensor(eps).reshape(1, -1))[0]


def _cpu(sd):
    return {k: v.detach().cpu().clone() for k, v in sd.items()}


def test_four_intersections_stream_matches_oracle():
    import hhrl_vae_amd.algorithm as A
    seconds = 60
    mcfg, wcfg = S.make_cfgs(manager_capacity=320, worker_capacity=160)      # learn_begin: 16 and 64 rows
    enc, mgr, wrk = A.Encoder(2, 2, seed=11), A.HATD3Triple(mcfg, seed=12), A.WorkerTD3(wcfg, seed=13)
    o_enc = OracleEncoder(_cpu(enc.state_dict()), 2)
    o_mgr = OracleNet("triple", mcfg, {n: _cpu(sd) for n, sd in mgr.state_dicts().items()})
    o_wrk = OracleNet("worker", wcfg, {n: _cpu(sd) for n, sd in wrk.state_dicts().items()})
    got, want = {}, {}
    c_got = S.run(enc, mgr, wrk, S.Stream(green=6), seconds, seed=5, trace=got)
    c_want = S.run(o_enc, o_mgr, o_wrk, S.Stream(green=6), seconds, seed=5, trace=want)
    for k in c_want:
        if not k.startswith("wall_"):
            assert c_got[k] == c_want[k], (k, c_got[k], c_want[k])
    # the cadence itself: 2 x 32 encodes per second, both levels learned, policies consulted
    assert c_got["encode"] == 2 * 32 * seconds
    assert c_got["learn_worker"] >= 40 and c_got["learn_manager"] >= 8
    assert c_got["choose_worker"] > 0 and c_got["choose_manager"] > 0 and c_got["decode"] > 0
    assert c_got["store_manager"] == mgr.pointer == o_mgr.pointer and c_got["store_worker"] == wrk.pointer == o_wrk.pointer
    for key in ("enc", "adv", "a_worker", "a_time", "a_phase", "a_goal"):
        assert len(got[key]) == len(want[key]) > 0, key
        g, w = np.stack(got[key]).astype(np.float64), np.stack(want[key]).astype(np.float64)
        # actions follow parameters that have taken Adam steps: their own floor is the parameters'
        scaled_close(key, g, w, rtol=1e-4, atol_rel=2e-6 if key in ("enc", "adv") else 2e-4)
    n = min(mgr.pointer, 320)
    scaled_close("manager replay", mgr.memory[:n], o_mgr.memory[:n], atol_rel=2e-4)
    n = min(wrk.pointer, 160)
    scaled_close("worker replay", wrk.memory[:n], o_wrk.memory[:n], atol_rel=2e-4)

    # Parameters at the end: every element within rtol 1e-4 + 2e-6 of the tensor's scale + what its own gradients
    # allow.  G_FLOOR is the relative uncertainty of a gradient between the two runs: their minibatches hold
    # actions and advised speeds that the runs computed themselves and that agree to 2e-4 of scale (above), and
    # a gradient inherits the relative spread of its inputs.  A target follows its network (tau = 0.005 per
    # policy step), so the network's bar bounds it too.  What that bar still checks: an element whose gradient sits
    # below ~3 G_FLOOR of its tensor's largest at most steps is allowed nearly all of Adam's reach and is in effect
    # unchecked at the end (it is checked through everything it feeds: the actions and replay rows above, the losses
    # and gradients of the kernel tests).  adam_path_close reports that share per network.  Elements whose reference
    # gradient is exactly zero at a step (weights behind units that are off for the whole minibatch: a third to a half
    # of these networks' elements) are held to NO movement at that step (zero_is_exact) instead of being counted as
    # gradients of unknown sign; what remains saturated is 4-9 % of the manager's elements over 14 / 7 steps and
    # 0.2-2 % of the worker's over 224 / 112, and MAX_SATURATED keeps it from growing unnoticed.
    G_FLOOR = 2e-4
    MAX_SATURATED = 0.15
    for net, onet, tag in ((mgr, o_mgr, "manager"), (wrk, o_wrk, "worker")):
        steps = onet.td3.total_it
        assert net.total_it == steps > 0 and len(onet.grad_log["critic"]) == steps
        sds, tgt = net.state_dicts(), net.target_state_dicts()
        np_ = lambda d: {k: v.detach().cpu().numpy() for k, v in d.items()}
        adam_path_close(tag + "/critic", np_(sds["critic"]), np_(onet.td3.critic), onet.grad_log["critic"],
                        onet.lr_log["critic"], g_floor=G_FLOOR, max_saturated=MAX_SATURATED, zero_is_exact=True)
        adam_path_close(tag + "/critic_target", np_(tgt["critic"]), np_(onet.td3.critic_target), onet.grad_log["critic"],
                        onet.lr_log["critic"], g_floor=G_FLOOR, max_saturated=MAX_SATURATED, zero_is_exact=True)
        for k, nme in enumerate(net.actor_names):
            assert len(onet.grad_log[nme]) == steps // 2
            adam_path_close(tag + "/" + nme, np_(sds[nme]), np_(onet.td3.actors[k]), onet.grad_log[nme], onet.lr_log[nme],
                            g_floor=G_FLOOR, max_saturated=MAX_SATURATED, zero_is_exact=True)
            adam_path_close(tag + "/" + nme + "_target", np_(tgt[nme]), np_(onet.td3.actor_targets[k]), onet.grad_log[nme],
                            onet.lr_log[nme], g_floor=G_FLOOR, max_saturated=MAX_SATURATED, zero_is_exact=True)


def test_four_intersections_batched_inference_same_values():
    """The batched inference calls (eight lanes per `encode_lanes`, the advised speeds of a light's head CAVs per `decode_host`, their actions per `policy`)
    feed the stream the same values as one call per lane / location: same call counts, encodings and
    advised speeds bit-identical while the two runs' inputs are (until the first update), and within 2e-6 of
    the tensor's scale throughout.  Actions, replay rows and parameters are held to the bars of the oracle
    comparison above: two runs of the TD3 updates differ in the last bits by themselves (the fused MLP
    backward adds weight-gradient rows with float atomics)."""
    import hhrl_vae_amd.algorithm as A
    mcfg, wcfg = S.make_cfgs(manager_capacity=320, worker_capacity=160)
    outs = []
    for batched in (False, True):
        enc, mgr, wrk = A.Encoder(2, 2, seed=11), A.HATD3Triple(mcfg, seed=12), A.WorkerTD3(wcfg, seed=13)
        tr = {}
        c = S.run(enc, mgr, wrk, S.Stream(green=6), 40, seed=7, trace=tr, batched=batched)
        torch.cuda.synchronize()
        outs.append((c, tr, mgr.memory.copy(), wrk.memory.copy(), mgr.state_dicts(), wrk.state_dicts(), wrk.total_it))
    (c0, t0, mm0, wm0, ms0, ws0, steps), (c1, t1, mm1, wm1, ms1, ws1, _) = outs
    assert c1["encode"] * 16 == c0["encode"] and c1["decode"] * 4 < c0["decode"] and c1["choose_worker"] * 2 < c0["choose_worker"] and c1["choose_manager"] * 2 < c0["choose_manager"]
    assert c0["learn_worker"] == c1["learn_worker"] > 0 and c0["learn_manager"] == c1["learn_manager"] > 0
    for key in t0:
        scaled_close(key, np.stack(t1[key]), np.stack(t0[key]), atol_rel=2e-6 if key in ("enc", "adv") else 2e-4)
    assert np.array_equal(np.stack(t0["enc"]), np.stack(t1["enc"]))          # lane observations do not depend on the agents
    assert np.array_equal(np.stack(t0["adv"][:64]), np.stack(t1["adv"][:64]))
    scaled_close("manager replay", mm1, mm0, atol_rel=2e-4)
    scaled_close("worker replay", wm1, wm0, atol_rel=2e-4)
    for a, b in ((ms0, ms1), (ws0, ws1)):
        for n in a:
            lr = 1e-3 if n == "critic" else 1e-4
            for k in a[n]:
                w = a[n][k].cpu().numpy()
                assert_close("%s/%s" % (n, k), b[n][k].cpu().numpy(), w, rtol=1e-4,
                             atol=2e-6 * max(float(np.max(np.abs(w))), 1e-30) + 5e-3 * lr * steps)
