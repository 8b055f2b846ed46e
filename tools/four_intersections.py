"""Four-intersection scenario: the call cadence of the reference's training loop on a synthetic
transition stream (no SUMO), driving any object set with the reference's class surface.

What the reference does per simulated second (train.py:183-204) with one holon that controls all four
intersections, `lane_agent=True`, time + phase + goal control (HATD3Triple) and a WorkerTD3 for the CAVs:

* manager, per light (agent.py:309-329): observation of the light appended to the T-step windows, 8 x
  `Encoder.encode(lane)`;
* manager, per light at the end of its green (agent.py:339-494): `choose_phase_action`,
  `choose_time_action`, 8 x `choose_goal` (or uniform random actions before `learn_begin`), exploration
  noise, 8 x `store_transition`, then `learn(worker_policy)` once the replay holds `learn_begin` rows;
* worker, per light (agent.py:748): 8 x `encode` again; per controlled head CAV (agent.py:787-853):
  2 x `decode` (advised speed here and at the next location), `choose_action`, `store_transition`;
  per light (agent.py:865-867): `learn()` once the replay holds `learn_begin` rows.

The environment is replaced by `Stream`: lane occupancies, light observations, CAV states and rewards are
pseudo-random functions of (seed, second, light, lane), so two runs with different implementations of the
classes see the same inputs as long as their outputs agree.  Every random draw the classes would make
themselves (minibatch rows, smoothing noise, actor order, decoder noise) and the agents' exploration noise
come from one generator owned by `run`, in a fixed order, and are handed to the classes.

    python tools/four_intersections.py [--seconds 300] [--json out.json]     (on a GPU box)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from collections import deque
from typing import Dict, List, Optional

import numpy as np

N_LIGHTS, N_LANES = 4, 8
LANE_LENGTH, MAX_SPEED = 300.0, 13.89          # environment.py:496-504 normalisers


class Stream:
    """Synthetic stand-in for the SUMO side: every query is a pure function of its arguments."""

    def __init__(self, seed: int = 3407, max_cars: int = 12, light_obs_dim: int = 13, cav_obs_dim: int = 7,
                 green: int = 12):
        self.seed, self.max_cars = seed, max_cars
        self.light_obs_dim, self.cav_obs_dim, self.green = light_obs_dim, cav_obs_dim, green

    def _rs(self, *key) -> np.random.RandomState:
        h = self.seed
        for k in key:
            h = (h * 1000003 + int(k) + 12345) % (2 ** 31 - 1)
        return np.random.RandomState(h)

    def lane_obs(self, t: int, light: int, lane: int) -> List[List[float]]:
        """[[1 - pos/300, v/13.89]] per car of the lane (environment.py:496-504); may be empty."""
        rs = self._rs(1, t, light, lane)
        n = int(rs.randint(0, self.max_cars + 1))
        x = np.sort(rs.uniform(0, 1, n))
        v = np.clip(rs.uniform(0, 1.08, n), 0, 1.08)
        return np.stack([x, v], 1).tolist() if n else []

    def light_obs(self, t: int, light: int) -> List[float]:
        return self._rs(2, t, light).uniform(0, 1, self.light_obs_dim).tolist()

    def lane_halt(self, t: int, light: int, lane: int) -> float:
        return float(self._rs(3, t, light, lane).randint(0, 6))

    def light_reward(self, t: int, light: int) -> float:
        return float(-self._rs(4, t, light).uniform(0, 1))

    def head_cav(self, t: int, light: int, lane: int) -> Optional[int]:
        """Id of the lane's controlled CAV or None; a CAV stays head of its lane for 6 s, then the lane
        is without one for 2 s."""
        phase = (t + 3 * lane + 5 * light) % 8
        return None if phase >= 6 else (((t + 3 * lane + 5 * light) // 8) * 64 + light * 8 + lane + 1)

    def cav_obs(self, t: int, cav: int) -> List[float]:
        return self._rs(5, t, cav).uniform(0, 1, self.cav_obs_dim).tolist()

    def cav_reward(self, t: int, cav: int) -> float:
        return float(-self._rs(6, t, cav).uniform(0, 1))

    def decision_point(self, t: int, light: int) -> bool:
        """End of a green: every `green` + 3 (yellow) + 2 (red) seconds, lights staggered."""
        return t >= 10 and (t + 4 * light) % (self.green + 5) == 0


def _draws(rng: np.random.RandomState, net, n_stored: int):
    """Minibatch rows (algorithm.py:286, 772, 1319: with replacement), smoothing noise per actor, actor
    order (algorithm.py:809), in that order."""
    B = net.batch_size
    idx = rng.choice(n_stored, size=B)
    noise = [rng.standard_normal((B, a)).astype(np.float32) for a in net.fam.a_dims]
    order = rng.permutation(len(net.fam.a_dims))
    return idx, noise, order


def run(encoder, manager, worker, stream: Stream, seconds: int, seed: int = 0, var: float = 0.3,
        trace: Optional[dict] = None, batched: bool = False) -> Dict[str, float]:
    """Drive the three objects through `seconds` simulated seconds; returns call counts and the wall
    time spent inside each kind of call.  `trace`, if given, collects the values that flow between the
    calls (encodings, advised speeds, actions) for comparison between implementations.  `batched` uses
    the calls the reference does not have: the eight lanes of a light in one `encode_lanes` call (shared
    by manager and worker, which encode the same lanes), the advised speeds of all head CAVs of a light in
    one `decode_host` call, their actions in one `policy` call and the eight lanes' goals at a decision point in
    one `choose_goal_batch` call; the values are the same."""
    rng = np.random.RandomState(seed)
    g2 = 2 * encoder.goal_dim
    T = manager.fam.streams[0].T
    a_dims = manager.fam.a_dims
    cnt = {k: 0 for k in ("encode", "decode", "choose_manager", "choose_worker", "store_manager", "store_worker",
                          "learn_manager", "learn_worker")}
    wall = {k: 0.0 for k in ("encode", "decode", "choose", "store", "learn_manager", "learn_worker")}
    tr = trace if trace is not None else {}
    for k in ("enc", "adv", "a_worker", "a_time", "a_phase", "a_goal"):
        tr.setdefault(k, [])

    def timed(key, fn, *a, **kw):
        t0 = time.perf_counter()
        out = fn(*a, **kw)
        wall[key] += time.perf_counter() - t0
        return out

    win = [deque(maxlen=T) for _ in range(N_LIGHTS)]                  # light observations, last T seconds
    enc_win = [deque(maxlen=T) for _ in range(N_LIGHTS)]              # lane encodings, last T seconds
    halt = [[] for _ in range(N_LIGHTS)]
    o_t, o_p, o_g, a_t, a_p, a_g = ([deque(maxlen=2) for _ in range(N_LIGHTS)] for _ in range(6))
    goal_state = [np.zeros((N_LANES, g2)) for _ in range(N_LIGHTS)]
    buf: Dict[int, dict] = {}                                          # per CAV: obs, goals, rewards
    n_m = lambda: min(manager.pointer, manager.memory_capacity)
    n_w = lambda: min(worker.pointer, worker.memory_capacity)

    for t in range(seconds):
        for L in range(N_LIGHTS):
            # ---- manager, every second (agent.py:309-329)
            obs = stream.light_obs(t, L)
            win[L].append(obs)
            lanes = [stream.lane_obs(t, L, l) for l in range(N_LANES)]
            if batched:
                enc = np.asarray(timed("encode", encoder.encode_lanes, lanes), dtype=np.float64)
                cnt["encode"] += 1
            else:
                enc = np.array([timed("encode", encoder.encode, lane) for lane in lanes])
                cnt["encode"] += N_LANES
            enc_win[L].append(enc)
            tr["enc"].append(enc)
            halt[L].append([stream.lane_halt(t, L, l) for l in range(N_LANES)])
            goal = None
            if stream.decision_point(t, L) and len(win[L]) == T:
                # ---- manager decision (agent.py:339-494)
                flat = np.array(win[L]).flatten().tolist()
                o_p[L].append(flat)
                o_t[L].append(flat)
                explore = manager.pointer < manager.learn_begin
                ap = rng.random_sample(a_dims[1]) * 2 - 1 if explore else timed("choose", manager.choose_phase_action, flat)
                ap = np.clip(rng.normal(0, var, size=ap.shape) + ap, -1, 1)
                one_hot = np.zeros_like(ap)
                one_hot[np.argmax(ap)] = 1.0
                a_p[L].append(one_hot)
                at = rng.random_sample(a_dims[0]) * 2 - 1 if explore else timed("choose", manager.choose_time_action, flat)
                at = np.clip(rng.normal(0, var, size=at.shape) + at, -1, 1)
                a_t[L].append(at)
                cnt["choose_manager"] += 0 if explore else 2
                tr["a_phase"].append(np.array(ap)); tr["a_time"].append(np.array(at))
                reward = stream.light_reward(t, L)
                lane_r = np.sum(np.array(halt[L]), 0) / len(halt[L])
                halt[L] = []
                g_obs, e = np.array(win[L]), np.array(enc_win[L])
                cur_o, cur_a = [], []
                for l in range(N_LANES):
                    lane_id = np.tile(np.eye(N_LANES)[l], (T, 1))
                    cur_o.append(np.concatenate((g_obs, e[:, l, :], lane_id), 1).flatten().tolist())
                goals = None
                if batched and not explore:                     # the eight lanes' goals in one call
                    goals = np.asarray(timed("choose", manager.choose_goal_batch, np.array(cur_o)), dtype=np.float64)
                    cnt["choose_manager"] += 1
                for l in range(N_LANES):
                    og = cur_o[l]
                    if explore:
                        ag = rng.random_sample(a_dims[2]) * 2 - 1
                    elif goals is not None:
                        ag = goals[l]
                    else:
                        ag = timed("choose", manager.choose_goal, og)
                        cnt["choose_manager"] += 1
                    ag = np.clip(rng.normal(0, var, size=ag.shape) + ag, -1, 1)
                    cur_a.append(ag)
                    reward = (2 * reward + lane_r[l]) / 3                       # agent.py:423 (running, as written)
                    if len(o_t[L]) >= 2 and len(o_g[L]) >= 1:
                        timed("store", manager.store_transition, o_t[L][-2], o_p[L][-2], o_g[L][-1][l],
                              a_t[L][-2], a_p[L][-2], a_g[L][-1][l], reward, o_t[L][-1], o_p[L][-1], og)
                        cnt["store_manager"] += 1
                o_g[L].append(cur_o)
                a_g[L].append(cur_a)
                goal = np.array(cur_a)
                tr["a_goal"].append(goal)
                if manager.pointer >= manager.learn_begin and n_m() > 0:
                    idx, noise, order = _draws(rng, manager, n_m())
                    timed("learn_manager", manager.learn, worker.policy, indices=idx, noise=noise, order=order)
                    cnt["learn_manager"] += 1
            # ---- worker, every second (agent.py:748-867)
            if batched:
                lane_state = enc.copy()
            else:
                lane_state = np.array([timed("encode", encoder.encode, stream.lane_obs(t, L, l)) for l in range(N_LANES)])
                cnt["encode"] += N_LANES
            if goal is not None:
                goal_state[L] = np.clip(lane_state + goal.reshape(N_LANES, -1), 0, 1)       # agent.py:766-772
            # pass 1, per head CAV in lane order: observation and every random draw of its step (decoder noise,
            # the uniform action while the replay fills, exploration noise) -- none depends on a network output
            cavs = []
            will_store = 0
            for l in range(N_LANES):
                cav = stream.head_cav(t, L, l)
                if cav is None:
                    continue
                ov = stream.cav_obs(t, cav) + lane_state[l].tolist()                         # use_raw_goal
                b = buf.setdefault(cav, {"obs": [], "goal": deque(maxlen=2), "reward": []})
                b["obs"].append(ov)
                eps = rng.standard_normal((2, encoder.goal_dim)).astype(np.float32)
                nxt = (ov[0] * LANE_LENGTH + ov[1] * MAX_SPEED) / LANE_LENGTH
                # the CAVs before this one have stored their transitions by the time it chooses (agent.py:820, 845)
                explore = worker.pointer + will_store < worker.learn_begin
                rand = rng.random_sample(worker.fam.a_dims[0]) * 2 - 1 if explore else None
                noise_v = rng.normal(0, var, size=(worker.fam.a_dims[0],))
                will_store += len(b["obs"]) >= 2
                cavs.append({"l": l, "cav": cav, "ov": ov, "b": b, "eps": eps, "nxt": nxt, "rand": rand, "noise": noise_v})
            # advised speeds at the CAV's location and at its next one (agent.py:809-811)
            if batched and cavs:
                goals2 = np.repeat(np.stack([goal_state[L][c["l"]] for c in cavs]), 2, axis=0)
                locs2 = np.array([[c["ov"][0]] if k == 0 else [c["nxt"]] for c in cavs for k in (0, 1)])
                both = timed("decode", encoder.decode_host, goals2, locs2, np.concatenate([c["eps"] for c in cavs]))
                cnt["decode"] += 1
                for i, c in enumerate(cavs):
                    c["adv"] = both[2 * i].tolist() + both[2 * i + 1].tolist()
            else:
                for c in cavs:
                    gs = goal_state[L][c["l"]]
                    adv0 = timed("decode", encoder.decode, gs, [c["ov"][0]], eps=c["eps"][0]).flatten().tolist()
                    adv1 = timed("decode", encoder.decode, gs, [c["nxt"]], eps=c["eps"][1]).flatten().tolist()
                    cnt["decode"] += 2
                    c["adv"] = adv0 + adv1
            for c in cavs:
                tr["adv"].append(np.array(c["adv"]))
                c["gv"] = goal_state[L][c["l"]].flatten().tolist() + c["adv"]
                c["b"]["goal"].append(c["gv"])
            # the policy for the CAVs that do not act at random (agent.py:820-826)
            acting = [c for c in cavs if c["rand"] is None]
            if batched and acting:
                acts = timed("choose", worker.policy, np.array([c["ov"] for c in acting]), np.array([c["gv"] for c in acting]))
                cnt["choose_worker"] += 1
                for c, a in zip(acting, np.asarray(acts)):
                    c["act"] = np.asarray(a, dtype=np.float64).reshape(-1)
            else:
                for c in acting:
                    c["act"] = timed("choose", worker.choose_action, [c["ov"]], c["gv"])
                    cnt["choose_worker"] += 1
            # pass 2: exploration noise, reward, transition (agent.py:824-851)
            for c in cavs:
                l, b, ov = c["l"], c["b"], c["ov"]
                av = c["rand"] if c["rand"] is not None else c["act"]
                av = np.clip(c["noise"] + av, -1, 1)
                tr["a_worker"].append(np.array(av))
                r = 0.5 * stream.cav_reward(t, c["cav"]) - 0.5 * float(np.sqrt(np.sum((goal_state[L][l] - lane_state[l]) ** 2)))
                b["reward"].append(r)
                if len(b["obs"]) >= 2:
                    timed("store", worker.store_transition, np.array(b["obs"][-2]).flatten(), ov[2], b["goal"][-2],
                          b["goal"][-1], r, np.array(b["obs"][-1]).flatten())
                    cnt["store_worker"] += 1
            if worker.pointer >= worker.learn_begin and n_w() > 0:
                idx, noise, order = _draws(rng, worker, n_w())
                timed("learn_worker", worker.learn, indices=idx, noise=noise, order=order)
                cnt["learn_worker"] += 1
        for cav in [c for c, b in buf.items() if len(b["obs"]) > 8]:
            del buf[cav]
    out = dict(cnt)
    out.update({"wall_" + k: v for k, v in wall.items()})
    return out


def make_cfgs(manager_capacity: int = 20000, worker_capacity: int = 120000):
    """Config dicts with the reference's default dimensions after the agents' in-place edits
    (configs.py:72-175, agent.py:225-238, 684-696): HATD3Triple with observations 13/13/25, T = 10,
    actions 1/4/4; WorkerTD3 with observation 11, goal 6, T = 1; Encoder(2, 2)."""
    mk = lambda o, a, h: {"obs_dim": o, "state_dim": 128, "act_dim": a, "T": 10, "hidden_dim": list(h)}
    m = {"use_opc": False, "time": mk(13, 1, (400, 300)), "phase": mk(13, 4, (400, 300)), "vehicle": mk(25, 4, (400, 300)),
         "tau": 0.005, "gamma": 0.95, "batch_size": 64, "memory_capacity": manager_capacity, "learn_start_ratio": 0.05,
         "actor_hidden_dim": [400, 300], "critic_hidden_dim": [512, 256], "actor_learning_rate": 1e-4,
         "critic_learning_rate": 1e-3, "actor_scheduler_step": 300, "critic_scheduler_step": 300}
    w = {"cav": {"obs_dim": 11, "state_dim": 32, "act_dim": 1, "T": 1, "hidden_dim": [128, 128]}, "goal_dim": 6,
         "hidden_dim": [128, 128], "batch_size": 64, "memory_capacity": worker_capacity, "learn_start_ratio": 0.4,
         "gamma": 0.9, "tau": 0.005, "actor_learning_rate": 1e-4, "critic_learning_rate": 1e-3,
         "actor_scheduler_step": 2000, "critic_scheduler_step": 1500}
    return m, w


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=int, default=300)
    ap.add_argument("--json", default="")
    ap.add_argument("--batched", action="store_true", help="encode_lanes / decode_batch instead of one call per lane / location")
    args = ap.parse_args()
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import hhrl_vae_amd.algorithm as A
    mcfg, wcfg = make_cfgs(manager_capacity=640, worker_capacity=320)     # learning starts within the first minute
    enc, mgr, wrk = A.Encoder(2, 2), A.HATD3Triple(mcfg), A.WorkerTD3(wcfg)
    st = Stream()
    run(enc, mgr, wrk, st, 60, batched=args.batched)                                             # warm-up: graphs captured, replay filled
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = run(enc, mgr, wrk, st, args.seconds, seed=1, batched=args.batched)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out.update({"workload": "four_intersections_synthetic_stream", "batched_inference": args.batched, "simulated_seconds": args.seconds, "wall_s": dt,
                "simulated_seconds_per_s": args.seconds / dt,
                "learning_path_wall_s": sum(v for k, v in out.items() if k.startswith("wall_"))})
    line = json.dumps(out)
    print(line)
    if args.json:
        with open(args.json, "w") as f:
            f.write(line + "\n")


if __name__ == "__main__":
    main()
