"""CPU oracle for the HHRL-VAE learning hot path.  TEST INFRASTRUCTURE ONLY.

This module is a plain, functional PyTorch-CPU (fp32) restatement of the
reference's learning arithmetic.  It exists to *check* the HIP path and to be
timed as the CPU baseline; it is never part of the shipped product.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it.

Pinning: every function below is checked against golden vectors recorded by
importing the reference's ``src/algorithm.py`` in the build container
(``tests/golden/make_golden.py`` wrote them, ``tests/test_oracle_golden.py``
replays them).  Oracle arithmetic version: torch CPU fp32.

All randomness is injected by the caller (``eps`` for the VAE, replay indices,
smoothing noise and actor update order for TD3), because the reference draws
it from global generators (algorithm.py:89, 286, 295, 817).

Parameters live in flat ``dict[str, Tensor]`` objects that use the reference's
``state_dict`` key names, so fixtures load without translation.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]


# --------------------------------------------------------------------------
# Layer primitives
# --------------------------------------------------------------------------
def linear(p: Params, name: str, x: Tensor) -> Tensor:
    """nn.Linear with weight (out, in): y = x W^T + b."""
    return F.linear(x, p[name + ".weight"], p[name + ".bias"])


# --------------------------------------------------------------------------
# VAE  (reference: algorithm.py:18-138)
# --------------------------------------------------------------------------
VAE_KEYS = [
    # order == Encoder.parameters() order == Adam state index order
    "encoder.phi.0", "encoder.phi.2", "encoder.rho.0", "encoder.rho.2",
    "encoder.to_mean", "encoder.to_logvar",
    "decoder.decoder.0", "decoder.decoder.2", "decoder.loc",
    "decoder.func.0", "decoder.func.2", "decoder.func.4",
]


def vae_param_shapes(goal_dim: int, car_info_dim: int = 2) -> List[Tuple[str, Tuple[int, ...]]]:
    """(name, shape) of every VAE parameter, in optimizer order.
    Widths follow Encoder.__init__ (algorithm.py:80: rho_dim=64, phi_dim=32,
    hidden_dim=128) and DeepSetDecoder.__init__ (algorithm.py:51-66)."""
    dims = {
        "encoder.phi.0": (16, car_info_dim), "encoder.phi.2": (32, 16),
        "encoder.rho.0": (128, 32), "encoder.rho.2": (64, 128),
        "encoder.to_mean": (goal_dim, 64), "encoder.to_logvar": (goal_dim, 64),
        "decoder.decoder.0": (128, goal_dim), "decoder.decoder.2": (32, 128),
        "decoder.loc": (32, 1),
        "decoder.func.0": (128, 64), "decoder.func.2": (32, 128), "decoder.func.4": (1, 32),
    }
    out = []
    for k in VAE_KEYS:
        out.append((k + ".weight", dims[k]))
        out.append((k + ".bias", (dims[k][0],)))
    return out


def deepset_encoder_forward(p: Params, x: Tensor) -> Tuple[Tensor, Tensor]:
    """DeepSetEncoder.forward, algorithm.py:40-47 (use_rho=True)."""
    h = F.relu(linear(p, "encoder.phi.0", x))
    h = linear(p, "encoder.phi.2", h)
    h = torch.sum(h, dim=-2)
    h = F.relu(linear(p, "encoder.rho.0", h))
    h = linear(p, "encoder.rho.2", h)
    return linear(p, "encoder.to_mean", h), linear(p, "encoder.to_logvar", h)


def deepset_decoder_forward(p: Params, g: Tensor, loc: Tensor) -> Tensor:
    """DeepSetDecoder.forward, algorithm.py:68-73."""
    g = F.relu(linear(p, "decoder.decoder.0", g))
    g = linear(p, "decoder.decoder.2", g)
    loc = linear(p, "decoder.loc", loc)
    x = torch.cat((g, loc), dim=-1)
    x = F.relu(linear(p, "decoder.func.0", x))
    x = F.relu(linear(p, "decoder.func.2", x))
    return torch.sigmoid(linear(p, "decoder.func.4", x))


def vae_loss(p: Params, data: Tensor, eps: Tensor,
             recon_count: Optional[int] = None) -> Tensor:
    """ELBO as written in Encoder.learn, algorithm.py:94-104.

    data: [B, N, 2] (col 0 = location, col 1 = speed); eps: [B, g].
    recon_count: divisor of the squared-error SUM.  None = B*N (MSELoss mean,
    algorithm.py:84,102).  A data-parallel shard passes the GLOBAL element
    count so that summing shard gradients reproduces the single-device step;
    the KL term is a plain sum (algorithm.py:103) and needs no rescaling.
    """
    loc = data[..., :1]
    true_v = data[..., 1:]
    mean, logvar = deepset_encoder_forward(p, data)
    z = mean + eps * torch.exp(0.5 * logvar)                    # :87-90
    decoded = deepset_decoder_forward(p, z.unsqueeze(-2).repeat(1, loc.shape[-2], 1), loc)
    if recon_count is None:
        recon = F.mse_loss(decoded, true_v)
    else:
        recon = torch.sum((decoded - true_v) ** 2) / float(recon_count)
    kl = -0.5 * torch.sum(1.0 + logvar - mean.pow(2) - logvar.exp())
    return recon + kl


# --------------------------------------------------------------------------
# Adam + StepLR  (torch.optim defaults; constructed at algorithm.py:83, 259-268)
# --------------------------------------------------------------------------
class Adam:
    """torch.optim.Adam with default hyper-parameters (betas (0.9, 0.999),
    eps 1e-8, no weight decay, no amsgrad), single-tensor update rule."""

    def __init__(self, params: Sequence[Tensor], lr: float):
        self.params = list(params)
        self.lr = float(lr)
        self.base_lr = float(lr)
        self.beta1, self.beta2, self.eps = 0.9, 0.999, 1e-8
        self.step_count = 0
        self.exp_avg = [torch.zeros_like(t) for t in self.params]
        self.exp_avg_sq = [torch.zeros_like(t) for t in self.params]

    @torch.no_grad()
    def step(self, grads: Sequence[Tensor]) -> None:
        self.step_count += 1
        t = self.step_count
        bc1 = 1.0 - self.beta1 ** t
        bc2 = 1.0 - self.beta2 ** t
        step_size = self.lr / bc1
        bc2_sqrt = math.sqrt(bc2)
        for p, g, m, v in zip(self.params, grads, self.exp_avg, self.exp_avg_sq):
            m.lerp_(g, 1.0 - self.beta1)
            v.mul_(self.beta2).addcmul_(g, g, value=1.0 - self.beta2)
            denom = (v.sqrt() / bc2_sqrt).add_(self.eps)
            p.addcdiv_(m, denom, value=-step_size)


class StepLR:
    """torch.optim.lr_scheduler.StepLR(step_size, gamma), chainable form: the optimizer's current lr is
    multiplied by gamma on every ``step_size``-th call of ``step()`` (so an lr restored from a checkpoint
    keeps decaying from where it was, algorithm.py:1375-1382)."""

    def __init__(self, opt: Adam, step_size: int, gamma: float = 0.5):
        self.opt, self.step_size, self.gamma, self.epoch = opt, int(step_size), gamma, 0

    def step(self) -> None:
        self.epoch += 1
        if self.epoch % self.step_size == 0:
            self.opt.lr *= self.gamma


class VAE:
    """Functional twin of ``Encoder`` (algorithm.py:76-138)."""

    def __init__(self, state: Params, goal_dim: int, lr: float = 1e-3, dtype=torch.float32):
        """dtype=torch.float64 gives a high-precision yardstick for judging
        fp32 summation noise; the oracle proper is fp32."""
        self.goal_dim = goal_dim
        self.dtype = dtype
        self.names = [n for n, _ in vae_param_shapes(goal_dim)]
        self.p: Params = {k: state[k].detach().clone().to(dtype).requires_grad_(True) for k in self.names}
        self.opt = Adam([self.p[k] for k in self.names], lr)
        self.grads: Params = {}

    def loss_and_grads(self, data: Tensor, eps: Tensor, recon_count=None):
        loss = vae_loss(self.p, data.to(self.dtype), eps.to(self.dtype), recon_count)
        g = torch.autograd.grad(loss, [self.p[k] for k in self.names])
        return loss.detach(), dict(zip(self.names, g))

    def learn(self, data, eps: Tensor) -> float:
        """Encoder.learn (algorithm.py:92-107) with the reparameterisation
        noise supplied by the caller."""
        data = torch.as_tensor(np.asarray(data, dtype=np.float32))
        loss, self.grads = self.loss_and_grads(data, eps)
        self.opt.step([self.grads[k] for k in self.names])
        return float(loss)

    @torch.no_grad()
    def encode(self, x) -> np.ndarray:
        """Encoder.encode (algorithm.py:109-116): one lane [N,2] -> mean||logvar."""
        if len(x) == 0:
            return np.zeros((self.goal_dim * 2))
        x = torch.as_tensor(np.asarray(x, dtype=np.float32))
        mean, logvar = deepset_encoder_forward(self.p, x)
        return np.concatenate((mean.numpy(), logvar.numpy()), axis=-1)

    @torch.no_grad()
    def decode(self, goal, loc, eps: Tensor) -> np.ndarray:
        """Encoder.decode (algorithm.py:121-127); eps replaces randn_like."""
        goal = np.asarray(goal, dtype=np.float32)
        mean = torch.as_tensor(goal[..., : self.goal_dim])
        logvar = torch.as_tensor(goal[..., self.goal_dim:])
        z = mean + eps * torch.exp(0.5 * logvar)
        x = torch.as_tensor(np.asarray(loc, dtype=np.float32))
        return deepset_decoder_forward(self.p, z, x).numpy()


def vae_chunk_dataset(dataset: Dict[str, list], batch_size: int = 64) -> List[list]:
    """Bucket-by-car-count chunking of VAE_train.py:55-56: every non-empty
    bucket is cut into consecutive chunks of <= batch_size lanes, buckets are
    visited in dict order, chunks concatenated in that order (unshuffled)."""
    chunks: List[list] = []
    for _, lanes in dataset.items():
        if len(lanes) == 0:
            continue
        for i in range(0, len(lanes), batch_size):
            chunks.append(lanes[i: i + batch_size])
    return chunks


# --------------------------------------------------------------------------
# BiLSTM front-end  (reference: algorithm.py:141-156)
# --------------------------------------------------------------------------
def _lstm_cell(x_t: Tensor, h: Tensor, c: Tensor, w_ih, w_hh, b_ih, b_hh):
    gates = F.linear(x_t, w_ih, b_ih) + F.linear(h, w_hh, b_hh)
    i, f, g, o = gates.chunk(4, dim=-1)            # PyTorch gate order i, f, g, o
    c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    h = torch.sigmoid(o) * torch.tanh(c)
    return h, c


def lstm_forward(p: Params, prefix: str, x: Tensor, T: int, obs_dim: int) -> Tensor:
    """LSTM.forward (algorithm.py:152-156): one-layer bidirectional LSTM,
    r_out[:, -1, :] then fc.  At the last time index the forward direction
    has consumed all T steps while the reverse direction has consumed only
    x[:, T-1] (its first step, from a zero state)."""
    x = x.reshape(-1, T, obs_dim)
    B = x.shape[0]
    L = prefix + "lstm_layer."
    hid = p[L + "weight_hh_l0"].shape[1]
    h = x.new_zeros(B, hid)
    c = x.new_zeros(B, hid)
    for t in range(T):
        h, c = _lstm_cell(x[:, t], h, c, p[L + "weight_ih_l0"], p[L + "weight_hh_l0"],
                          p[L + "bias_ih_l0"], p[L + "bias_hh_l0"])
    hr, _ = _lstm_cell(x[:, T - 1], x.new_zeros(B, hid), x.new_zeros(B, hid),
                       p[L + "weight_ih_l0_reverse"], p[L + "weight_hh_l0_reverse"],
                       p[L + "bias_ih_l0_reverse"], p[L + "bias_hh_l0_reverse"])
    return linear(p, prefix + "fc", torch.cat((h, hr), dim=-1))


def lstm_param_shapes(prefix: str, obs_dim: int, hid: int):
    L = prefix + "lstm_layer."
    out = []
    for suf in ("", "_reverse"):
        out += [(L + "weight_ih_l0" + suf, (4 * hid, obs_dim)), (L + "weight_hh_l0" + suf, (4 * hid, hid)),
                (L + "bias_ih_l0" + suf, (4 * hid,)), (L + "bias_hh_l0" + suf, (4 * hid,))]
    out += [(prefix + "fc.weight", (hid, 2 * hid)), (prefix + "fc.bias", (hid,))]
    return out


# --------------------------------------------------------------------------
# TD3 family description
# --------------------------------------------------------------------------
@dataclass
class Stream:
    """One observation stream: per-step width o, window T, LSTM width s."""
    o: int
    T: int
    s: int

    @property
    def width(self) -> int:          # flattened replay width of this stream
        return self.o * self.T

    @property
    def feat(self) -> int:           # width after the (optional) LSTM
        return self.s if self.T > 1 else self.o


@dataclass
class Family:
    """Shape description shared by all TD3 variants of algorithm.py.

    kind: 'single' (TD3Single :235), 'double' (HATD3Double :409 / HATD3 :567),
    'triple' (HATD3Triple :679), 'manager' (ManagerTD3 :915), 'worker' (WorkerTD3 :1261).
    """
    kind: str
    streams: List[Stream]
    a_dims: List[int]
    actor_hidden: List[int]
    critic_hidden: List[int]
    goal_dim: int = 0
    gamma: float = 0.95
    tau: float = 0.005
    batch_size: int = 64
    actor_lr: float = 1e-4
    critic_lr: float = 1e-3
    actor_sched: int = 300
    critic_sched: int = 300
    policy_freq: int = 2

    @property
    def obs_dim(self) -> int:
        return sum(s.width for s in self.streams)

    @property
    def act_dim(self) -> int:
        return sum(self.a_dims)

    @property
    def row_width(self) -> int:
        return 2 * (self.obs_dim + self.goal_dim) + self.act_dim + 1

    def actor_names(self) -> List[str]:
        return ["actor"] if len(self.a_dims) == 1 else ["actor%d" % (i + 1) for i in range(len(self.a_dims))]

    def critic_lstm_prefixes(self) -> List[str]:
        if len(self.streams) == 1:
            return ["lstm."]
        return ["lstm%d." % (i + 1) for i in range(len(self.streams))]


def family_from_cfg(kind: str, cfg: dict, task: Optional[str] = None) -> Family:
    """Read the dimensions the reference constructors read
    (algorithm.py:236-271, 410-463, 568-606, 680-748, 953-991, 1262-1297)."""
    def st(key):
        return Stream(cfg[key]["obs_dim"], cfg[key]["T"], cfg[key]["state_dim"])
    common = dict(gamma=cfg["gamma"], tau=cfg["tau"], batch_size=cfg["batch_size"],
                  actor_lr=cfg["actor_learning_rate"], critic_lr=cfg["critic_learning_rate"],
                  actor_sched=cfg["actor_scheduler_step"], critic_sched=cfg["critic_scheduler_step"])
    if kind == "single":
        return Family(kind, [st(task)], [cfg[task]["act_dim"]], cfg[task]["hidden_dim"], cfg[task]["hidden_dim"], **common)
    if kind == "double":        # task = light_opt ('time' | 'phase'); second stream is 'vehicle'
        keys = [task, "vehicle"]
    elif kind == "tp":          # HATD3: 'time' + 'phase'
        kind, keys = "double", ["time", "phase"]
    elif kind == "triple":
        keys = ["time", "phase", "vehicle"]
    elif kind == "manager":
        return Family(kind, [st("vehicle")], [cfg["vehicle"]["act_dim"]], cfg["vehicle"]["hidden_dim"],
                      cfg["vehicle"]["hidden_dim"], **common)
    elif kind == "worker":
        return Family(kind, [st("cav")], [cfg["cav"]["act_dim"]], cfg["hidden_dim"], cfg["hidden_dim"],
                      goal_dim=cfg["goal_dim"], **common)
    else:
        raise ValueError(kind)
    return Family(kind, [st(k) for k in keys], [cfg[k]["act_dim"] for k in keys],
                  cfg["actor_hidden_dim"], cfg["critic_hidden_dim"], **common)


def actor_param_shapes(fam: Family, k: int):
    """Parameter (name, shape) list of actor k in nn.Module registration
    order (Actor algorithm.py:159-172; WorkerActor :1187-1200)."""
    s = fam.streams[k]
    out = []
    if s.T > 1:
        out += lstm_param_shapes("lstm.", s.o, s.s)
    d_in = s.feat + fam.goal_dim
    h0, h1 = fam.actor_hidden
    for name, shp in (("l1", (h0, d_in)), ("l2", (h1, h0)), ("l3", (fam.a_dims[k], h1))):
        out += [(name + ".weight", shp), (name + ".bias", (shp[0],))]
    return out


def critic_param_shapes(fam: Family):
    """Critic parameters in registration order (CriticSingle :188-206,
    CriticDouble :352-375, CriticTriple :614-642, ManagerCritic :867-885,
    WorkerCritic :1214-1232)."""
    out = []
    for s, pre in zip(fam.streams, fam.critic_lstm_prefixes()):
        if s.T > 1:
            out += lstm_param_shapes(pre, s.o, s.s)
    d_in = sum(s.feat for s in fam.streams) + fam.goal_dim + fam.act_dim
    h0, h1 = fam.critic_hidden
    for name, shp in (("l1", (h0, d_in)), ("l2", (h1, h0)), ("l3", (1, h1)),
                      ("l4", (h0, d_in)), ("l5", (h1, h0)), ("l6", (1, h1))):
        out += [(name + ".weight", shp), (name + ".bias", (shp[0],))]
    return out


def actor_forward(p: Params, fam: Family, k: int, obs: Tensor, goal: Optional[Tensor] = None) -> Tensor:
    """Actor.forward (algorithm.py:174-180) / WorkerActor.forward (:1202-1211); max_action = 1."""
    s = fam.streams[k]
    x = lstm_forward(p, "lstm.", obs, s.T, s.o) if s.T > 1 else obs.reshape(obs.shape[0], -1)
    if goal is not None:
        x = torch.cat([x, goal], 1)
    x = F.relu(linear(p, "l1", x))
    x = F.relu(linear(p, "l2", x))
    return torch.tanh(linear(p, "l3", x))


def critic_forward(p: Params, fam: Family, obs: Sequence[Tensor], action: Tensor,
                   goal: Optional[Tensor] = None, q1_only: bool = False):
    """Twin-Q critic forward / Q1 (algorithm.py:208-232, 377-405, 644-676, 887-911, 1234-1258)."""
    feats = []
    for s, pre, o in zip(fam.streams, fam.critic_lstm_prefixes(), obs):
        feats.append(lstm_forward(p, pre, o, s.T, s.o) if s.T > 1 else o)
    if goal is not None:
        feats.append(goal)
    feats.append(action)
    xu = torch.cat(feats, 1)
    x1 = F.relu(linear(p, "l1", xu))
    x1 = F.relu(linear(p, "l2", x1))
    x1 = linear(p, "l3", x1)
    if q1_only:
        return x1
    x2 = F.relu(linear(p, "l4", xu))
    x2 = F.relu(linear(p, "l5", x2))
    x2 = linear(p, "l6", x2)
    return x1, x2


@dataclass
class Batch:
    """One sampled minibatch, already split the way the reference slices it."""
    obs: List[Tensor]
    act: List[Tensor]
    reward: Tensor
    next_obs: List[Tensor]
    goal: Optional[Tensor] = None
    next_goal: Optional[Tensor] = None


def split_rows(fam: Family, rows: np.ndarray) -> Batch:
    """Slice replay rows into tensors (algorithm.py:289-292, 485-491, 774-783, 1322-1327).

    Row layout: obs streams | actions | [goal | next_goal] | reward | next obs streams.
    """
    rows = np.asarray(rows)
    ft = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32))
    pos = 0
    obs = []
    for s in fam.streams:
        obs.append(ft(rows[:, pos: pos + s.width])); pos += s.width
    act = []
    for a in fam.a_dims:
        act.append(ft(rows[:, pos: pos + a])); pos += a
    goal = next_goal = None
    if fam.goal_dim:
        goal = ft(rows[:, pos: pos + fam.goal_dim]); pos += fam.goal_dim
        next_goal = ft(rows[:, pos: pos + fam.goal_dim]); pos += fam.goal_dim
    reward = ft(rows[:, pos: pos + 1]); pos += 1
    nobs = []
    for s in fam.streams:
        nobs.append(ft(rows[:, pos: pos + s.width])); pos += s.width
    assert pos == rows.shape[1] == fam.row_width
    return Batch(obs, act, reward, nobs, goal, next_goal)


class TD3:
    """Functional twin of the reference's TD3 variants.

    ``learn(batch, noise, order)`` follows WorkerTD3.learn (algorithm.py:1316-1366),
    TD3Single.learn (:283-328), ManagerTD3.learn (:1109-1168; with use_opc the caller relabels the
    batch's actions with `off_policy_corrections` first),
    HATD3Double.learn (:479-542) and HATD3Triple.learn (:768-838).
    """

    def __init__(self, fam: Family, actor_states: Sequence[Params], critic_state: Params):
        self.fam = fam
        rg = lambda sd, names: {k: sd[k].detach().clone().float().requires_grad_(True) for k in names}
        self.actor_names = [[n for n, _ in actor_param_shapes(fam, k)] for k in range(len(fam.a_dims))]
        self.critic_names = [n for n, _ in critic_param_shapes(fam)]
        self.actors = [rg(sd, names) for sd, names in zip(actor_states, self.actor_names)]
        self.critic = rg(critic_state, self.critic_names)
        cp = lambda d: {k: v.detach().clone() for k, v in d.items()}
        self.actor_targets = [cp(a) for a in self.actors]
        self.critic_target = cp(self.critic)
        self.actor_opts = [Adam([a[k] for k in n], fam.actor_lr) for a, n in zip(self.actors, self.actor_names)]
        self.critic_opt = Adam([self.critic[k] for k in self.critic_names], fam.critic_lr)
        self.actor_scheds = [StepLR(o, fam.actor_sched) for o in self.actor_opts]
        self.critic_sched = StepLR(self.critic_opt, fam.critic_sched)
        self.total_it = 0
        self.last = {}

    # -- pieces -----------------------------------------------------------
    def target_q(self, b: Batch, noise: Sequence[Tensor]) -> Tensor:
        fam = self.fam
        with torch.no_grad():
            nas = []
            for k, (pt, eps) in enumerate(zip(self.actor_targets, noise)):
                n = (eps * 0.2).clamp(-0.5, 0.5)
                nas.append((actor_forward(pt, fam, k, b.next_obs[k], b.next_goal) + n).clamp(-1, 1))
            q1, q2 = critic_forward(self.critic_target, fam, b.next_obs, torch.cat(nas, 1), b.next_goal)
            return b.reward + fam.gamma * torch.min(q1, q2)

    def critic_loss(self, b: Batch, y: Tensor) -> Tensor:
        q1, q2 = critic_forward(self.critic, self.fam, b.obs, torch.cat(b.act, 1), b.goal)
        return F.mse_loss(q1, y) + F.mse_loss(q2, y)

    def actor_loss(self, b: Batch) -> Tensor:
        fam = self.fam
        joint = torch.cat([actor_forward(a, fam, j, b.obs[j], b.goal) for j, a in enumerate(self.actors)], 1)
        return -critic_forward(self.critic, fam, b.obs, joint, b.goal, q1_only=True).mean()

    @torch.no_grad()
    def _polyak(self, net: Params, tgt: Params) -> None:
        tau = self.fam.tau
        for k in net:
            tgt[k].copy_(tau * net[k] + (1 - tau) * tgt[k])

    # -- the update -------------------------------------------------------
    def learn(self, b: Batch, noise: Sequence[Tensor], order: Optional[Sequence[int]] = None) -> dict:
        fam = self.fam
        self.total_it += 1
        y = self.target_q(b, noise)
        closs = self.critic_loss(b, y)
        cg = torch.autograd.grad(closs, [self.critic[k] for k in self.critic_names])
        self.critic_opt.step(cg)
        self.critic_sched.step()
        out = {"critic_loss": float(closs.detach()), "critic_grads": dict(zip(self.critic_names, cg)), "target_q": y}
        if self.total_it % fam.policy_freq == 0:
            n_act = len(self.actors)
            if order is None:
                order = list(range(n_act))
            out["actor_loss"], out["actor_grads"] = {}, {}
            for i in order:
                i = int(i)
                aloss = self.actor_loss(b)
                ag = torch.autograd.grad(aloss, [self.actors[i][k] for k in self.actor_names[i]])
                self.actor_opts[i].step(ag)
                out["actor_loss"][i] = float(aloss.detach())
                out["actor_grads"][i] = dict(zip(self.actor_names[i], ag))
            self._polyak(self.critic, self.critic_target)
            for a, t in zip(self.actors, self.actor_targets):
                self._polyak(a, t)
            for s in self.actor_scheds:
                s.step()
        self.last = out
        return out

    @torch.no_grad()
    def choose(self, k: int, obs, goal=None) -> np.ndarray:
        """choose_action / choose_*_action / choose_goal (algorithm.py:279-281,
        471-477, 756-766, 1003-1005, 1305-1308): batch-1 actor forward."""
        s = self.fam.streams[k]
        o = torch.as_tensor(np.asarray(obs, dtype=np.float32)).reshape(1, -1, s.o)
        g = None if goal is None else torch.as_tensor(np.asarray(goal, dtype=np.float32)).reshape(1, -1)
        o2 = o if s.T > 1 else o.reshape(1, -1)
        return actor_forward(self.actors[k], self.fam, k, o2, g).numpy().flatten()

    @torch.no_grad()
    def policy(self, obs, goal) -> np.ndarray:
        """WorkerTD3.policy (algorithm.py:1310-1314): actor 0 on n rows of (observation, goal)."""
        o = torch.as_tensor(np.asarray(obs, dtype=np.float32))
        g = torch.as_tensor(np.asarray(goal, dtype=np.float32))
        return actor_forward(self.actors[0], self.fam, 0, o, g).numpy()


def off_policy_corrections(worker_policy, sgoals, wstates, wactions, candidate_goals: int = 8) -> np.ndarray:
    """ManagerTD3.off_policy_corrections (algorithm.py:1007-1107), one lane and one candidate at a time.

    sgoals [B, L] stored manager goals; wstates [B, S, L, wo] / wactions [B, S, L, wa] what the workers
    saw and did while that goal was active.  Per lane the candidates are the stored goal, the change of
    the goal feature (last observation feature) over the S steps, and `candidate_goals` normal draws
    around that change (sigma 0.5, clipped to [-1, 1], numpy's global generator).  The goal a worker
    sees at step s under candidate c is c + feature_0 - feature_s (:1073).  Steps whose first
    observation feature is -1 take action -1 (:1084-1086; `np.any` is applied to the row NUMBERS there,
    so a lane whose only such row is row 0 is left alone -- reproduced).  Score = -0.5 * sum over steps
    of the squared action error (:1096); the best-scoring candidate replaces the stored goal.
    """
    wstates, wactions, sgoals = np.asarray(wstates), np.asarray(wactions), np.asarray(sgoals)
    B, S, L, wo = wstates.shape
    wa = wactions.shape[-1]
    out = np.zeros((B, L))
    for lane in range(L):
        st = wstates[:, :, lane, :]                                   # [B, S, wo]
        ac = wactions[:, :, lane, :].reshape(B * S, wa)
        feat = st[:, :, -1:]                                          # the goal feature, [B, S, 1]
        change = (st[:, -1] - st[:, 0])[:, None, -1:]                 # [B, 1, 1]
        draws = np.random.normal(loc=change, scale=.5, size=(B, candidate_goals, 1)).clip(-1, 1)
        cands = np.concatenate([sgoals[:, lane][:, None, None], change, draws], axis=1)      # [B, C, 1]
        rows = st.reshape(B * S, wo)
        empty = np.where(rows[:, 0] == -1)
        score = np.zeros((B, cands.shape[1]))
        for c in range(cands.shape[1]):
            seen = (cands[:, c] + feat[:, 0])[:, None] - feat         # [B, S, 1]
            a = np.array(worker_policy(rows, seen.reshape(B * S, 1)), dtype=np.float64)
            if np.any(empty):
                a[empty] = -1
            err = a - ac
            err = np.where(err != -np.inf, err, 0).reshape(B, S, wa)
            score[:, c] = -0.5 * np.sum(np.linalg.norm(err, axis=-1) ** 2, axis=-1)
        out[:, lane] = cands[np.arange(B), np.argmax(score, axis=-1), 0]
    return out
