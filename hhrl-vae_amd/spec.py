"""Shape descriptions of the TD3 variants, read from the reference's config
dictionaries exactly where its constructors read them."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple

Shape = Tuple[int, ...]


@dataclass
class Stream:
    """One observation stream: per-step width o, window length T, LSTM width s."""
    o: int
    T: int
    s: int

    @property
    def width(self) -> int:
        return self.o * self.T

    @property
    def feat(self) -> int:
        return self.s if self.T > 1 else self.o


@dataclass
class Family:
    kind: str                     # single | double | triple | manager | worker
    streams: List[Stream]
    a_dims: List[int]
    actor_hidden: List[int]
    critic_hidden: List[int]
    goal_dim: int
    gamma: float
    tau: float
    batch_size: int
    memory_capacity: int
    learn_start_ratio: float
    actor_lr: float
    critic_lr: float
    actor_sched: int
    critic_sched: int
    policy_freq: int = 2          # algorithm.py:270, 462, 747, 990, 1296

    @property
    def obs_dim(self) -> int:
        return sum(s.width for s in self.streams)

    @property
    def act_dim(self) -> int:
        return sum(self.a_dims)

    @property
    def row_width(self) -> int:
        """obs | actions | [goal | next goal] | reward | next obs
        (algorithm.py:252, 435, 709, 1276)."""
        return 2 * (self.obs_dim + self.goal_dim) + self.act_dim + 1

    def actor_names(self) -> List[str]:
        return ["actor"] if len(self.a_dims) == 1 else ["actor%d" % (i + 1) for i in range(len(self.a_dims))]

    def critic_lstm_prefixes(self) -> List[str]:
        return ["lstm."] if len(self.streams) == 1 else ["lstm%d." % (i + 1) for i in range(len(self.streams))]

    @property
    def critic_in(self) -> int:
        return sum(s.feat for s in self.streams) + self.goal_dim + self.act_dim


def family_from_cfg(kind: str, cfg: dict, task: Optional[str] = None) -> Family:
    """kind/task -> dimensions, following TD3Single.__init__ (algorithm.py:236-271),
    HATD3Double (:410-463), HATD3 (:568-606), HATD3Triple (:680-748),
    ManagerTD3 (:953-991) and WorkerTD3 (:1262-1297)."""
    def st(key: str) -> Stream:
        return Stream(int(cfg[key]["obs_dim"]), int(cfg[key]["T"]), int(cfg[key]["state_dim"]))

    common = dict(gamma=float(cfg["gamma"]), tau=float(cfg["tau"]), batch_size=int(cfg["batch_size"]),
                  memory_capacity=int(cfg["memory_capacity"]), learn_start_ratio=float(cfg["learn_start_ratio"]),
                  actor_lr=float(cfg["actor_learning_rate"]), critic_lr=float(cfg["critic_learning_rate"]),
                  actor_sched=int(cfg["actor_scheduler_step"]), critic_sched=int(cfg["critic_scheduler_step"]))
    if kind == "single":
        h = list(cfg[task]["hidden_dim"])
        return Family(kind, [st(task)], [int(cfg[task]["act_dim"])], h, h, 0, **common)
    if kind == "manager":
        h = list(cfg["vehicle"]["hidden_dim"])
        return Family(kind, [st("vehicle")], [int(cfg["vehicle"]["act_dim"])], h, h, 0, **common)
    if kind == "worker":
        h = list(cfg["hidden_dim"])
        return Family(kind, [st("cav")], [int(cfg["cav"]["act_dim"])], h, h, int(cfg["goal_dim"]), **common)
    if kind == "double":
        keys = [task, "vehicle"]
    elif kind == "tp":
        kind, keys = "double", ["time", "phase"]
    elif kind == "triple":
        keys = ["time", "phase", "vehicle"]
    else:
        raise ValueError("unknown TD3 family %r" % kind)
    return Family(kind, [st(k) for k in keys], [int(cfg[k]["act_dim"]) for k in keys],
                  list(cfg["actor_hidden_dim"]), list(cfg["critic_hidden_dim"]), 0, **common)


def lstm_shapes(prefix: str, obs_dim: int, hid: int) -> List[Tuple[str, Shape]]:
    """nn.LSTM(bidirectional) + fc parameter order (algorithm.py:148-150)."""
    L = prefix + "lstm_layer."
    out: List[Tuple[str, Shape]] = []
    for suf in ("", "_reverse"):
        out += [(L + "weight_ih_l0" + suf, (4 * hid, obs_dim)), (L + "weight_hh_l0" + suf, (4 * hid, hid)),
                (L + "bias_ih_l0" + suf, (4 * hid,)), (L + "bias_hh_l0" + suf, (4 * hid,))]
    out += [(prefix + "fc.weight", (hid, 2 * hid)), (prefix + "fc.bias", (hid,))]
    return out


def actor_shapes(fam: Family, k: int) -> List[Tuple[str, Shape]]:
    """Actor (algorithm.py:159-172) / WorkerActor (:1187-1200) in registration order."""
    s = fam.streams[k]
    out: List[Tuple[str, Shape]] = []
    if s.T > 1:
        out += lstm_shapes("lstm.", s.o, s.s)
    d_in = s.feat + fam.goal_dim
    h0, h1 = fam.actor_hidden
    for name, shp in (("l1", (h0, d_in)), ("l2", (h1, h0)), ("l3", (fam.a_dims[k], h1))):
        out += [(name + ".weight", shp), (name + ".bias", (shp[0],))]
    return out


def critic_shapes(fam: Family) -> List[Tuple[str, Shape]]:
    """Twin critics (algorithm.py:188-206, 352-375, 614-642, 867-885, 1214-1232)."""
    out: List[Tuple[str, Shape]] = []
    for s, pre in zip(fam.streams, fam.critic_lstm_prefixes()):
        if s.T > 1:
            out += lstm_shapes(pre, s.o, s.s)
    h0, h1 = fam.critic_hidden
    for name, shp in (("l1", (h0, fam.critic_in)), ("l2", (h1, h0)), ("l3", (1, h1)),
                      ("l4", (h0, fam.critic_in)), ("l5", (h1, h0)), ("l6", (1, h1))):
        out += [(name + ".weight", shp), (name + ".bias", (shp[0],))]
    return out
