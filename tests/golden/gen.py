"""Deterministic input generators shared by the golden-vector writer and the
tests.  Nothing here touches the reference: weights, replay rows and VAE
batches are drawn from ``numpy.random.RandomState`` (a frozen legacy stream),
so a fixture can name a seed instead of storing megabytes of random floats.
"""
from __future__ import annotations

import zlib
from typing import Dict, List, Sequence, Tuple

import numpy as np


def _seed(*parts) -> int:
    return zlib.crc32("/".join(str(p) for p in parts).encode()) & 0x7FFFFFFF


def gen_state(shapes: Sequence[Tuple[str, Tuple[int, ...]]], seed: int) -> Dict[str, np.ndarray]:
    """U(-k, k) with k = 1/sqrt(fan_in) per tensor (fan_in of the matching
    weight for biases), drawn tensor by tensor in the given order."""
    rs = np.random.RandomState(seed)
    out: Dict[str, np.ndarray] = {}
    last_fan = 1
    for name, shp in shapes:
        if len(shp) == 2:
            last_fan = shp[1]
            if "weight_hh" in name or "weight_ih" in name:
                last_fan = shp[0] // 4          # LSTM: 1/sqrt(hidden)
        k = 1.0 / np.sqrt(max(last_fan, 1))
        out[name] = rs.uniform(-k, k, size=shp).astype(np.float32)
    return out


def gen_rows(n_rows: int, width: int, seed: int) -> np.ndarray:
    """Replay rows ~ U(0, 1), float64 like the reference's numpy memory."""
    return np.random.RandomState(seed).uniform(0.0, 1.0, size=(n_rows, width))


def gen_lanes(B: int, N: int, seed: int) -> np.ndarray:
    """VAE batch [B, N, 2]: x ~ U(0,1), v ~ U(0, 1.08) (route files allow a
    max speed slightly above the normaliser)."""
    rs = np.random.RandomState(seed)
    x = rs.uniform(0.0, 1.0, size=(B, N, 1))
    v = rs.uniform(0.0, 1.08, size=(B, N, 1))
    return np.concatenate([x, v], axis=-1).astype(np.float32)


def digest(a: np.ndarray, n_samples: int = 16) -> np.ndarray:
    """Compact fingerprint of a tensor: [sum, sum|.|, n_samples strided values]
    (float64).  Used where storing the full tensor would bloat the fixtures."""
    f = np.asarray(a, dtype=np.float64).ravel()
    idx = np.linspace(0, f.size - 1, num=min(f.size, n_samples)).astype(np.int64)
    pad = np.zeros(n_samples - idx.size)
    return np.concatenate([[f.sum(), np.abs(f).sum()], f[idx], pad])


# ---------------------------------------------------------------- configs
def light_cfg(o=(13, 13, 25), T=10, s=128, a=(1, 4, 4), actor_h=(400, 300), critic_h=(512, 256),
              veh_h=(400, 300), batch=64, cap=256, a_sched=300, c_sched=300) -> dict:
    """A manager-side config dict with the keys the reference constructors
    read (shape of configs.py:72-132 after the agent's in-place edits)."""
    mk = lambda i, h: {"obs_dim": o[i], "state_dim": s, "act_dim": a[i], "T": T, "hidden_dim": list(h)}
    return {
        "use_opc": False,
        "time": mk(0, actor_h), "phase": mk(1, actor_h), "vehicle": mk(2, veh_h),
        "tau": 0.005, "gamma": 0.95, "batch_size": batch, "memory_capacity": cap,
        "learn_start_ratio": 0.05,
        "actor_hidden_dim": list(actor_h), "critic_hidden_dim": list(critic_h),
        "actor_learning_rate": 1e-4, "critic_learning_rate": 1e-3,
        "actor_scheduler_step": a_sched, "critic_scheduler_step": c_sched,
    }


def cav_cfg(o=11, T=1, s=32, a=1, g=6, hidden=(128, 128), batch=64, cap=256, a_sched=2000, c_sched=1500) -> dict:
    """A worker-side config dict (shape of configs.py:135-175 after
    agent.py:694-696 added the lane-state width to obs_dim and goal_dim)."""
    return {
        "cav": {"obs_dim": o, "state_dim": s, "act_dim": a, "T": T, "hidden_dim": list(hidden)},
        "goal_dim": g, "hidden_dim": list(hidden),
        "batch_size": batch, "memory_capacity": cap, "learn_start_ratio": 0.4,
        "gamma": 0.9, "tau": 0.005,
        "actor_learning_rate": 1e-4, "critic_learning_rate": 1e-3,
        "actor_scheduler_step": a_sched, "critic_scheduler_step": c_sched,
    }


# name -> (kind, task, cfg-kwargs builder, n_rows, n_calls, mode)
#   mode 'full'   : every tensor stored
#   mode 'digest' : weights / rows regenerated from seeds, outputs fingerprinted
TD3_CASES = {
    "worker_small": ("worker", None, lambda: cav_cfg(o=5, g=3, hidden=(16, 12), batch=8, cap=48, a_sched=1, c_sched=2), 40, 4, "full"),
    "worker_lstm_small": ("worker", None, lambda: cav_cfg(o=4, T=3, s=8, g=2, hidden=(12, 10), batch=6, cap=48, a_sched=1, c_sched=2), 30, 4, "full"),
    "single_cav_small": ("single", "cav", lambda: cav_cfg(o=7, g=0, hidden=(16, 12), batch=8, cap=48, a_sched=1, c_sched=2), 40, 4, "full"),
    "single_time_small": ("single", "time", lambda: light_cfg(o=(5, 5, 7), T=4, s=8, a=(1, 2, 3), actor_h=(12, 10), critic_h=(14, 12), veh_h=(12, 10), batch=8, cap=48, a_sched=1, c_sched=2), 40, 4, "full"),
    "manager_small": ("manager", None, lambda: light_cfg(o=(5, 5, 7), T=4, s=8, a=(1, 2, 3), actor_h=(12, 10), critic_h=(14, 12), veh_h=(12, 10), batch=8, cap=48, a_sched=1, c_sched=2), 40, 4, "full"),
    "double_small": ("double", "time", lambda: light_cfg(o=(5, 5, 7), T=4, s=8, a=(1, 2, 3), actor_h=(12, 10), critic_h=(14, 12), veh_h=(12, 10), batch=8, cap=48, a_sched=1, c_sched=2), 40, 4, "full"),
    "double_phase_small": ("double", "phase", lambda: light_cfg(o=(5, 6, 7), T=3, s=8, a=(1, 2, 3), actor_h=(12, 10), critic_h=(14, 12), veh_h=(12, 10), batch=8, cap=48, a_sched=2, c_sched=3), 40, 4, "full"),
    "tp_small": ("tp", None, lambda: light_cfg(o=(5, 6, 7), T=4, s=8, a=(1, 2, 3), actor_h=(12, 10), critic_h=(14, 12), veh_h=(12, 10), batch=8, cap=48, a_sched=1, c_sched=2), 40, 4, "full"),
    "triple_small": ("triple", None, lambda: light_cfg(o=(5, 6, 7), T=4, s=8, a=(1, 2, 3), actor_h=(12, 10), critic_h=(14, 12), veh_h=(12, 10), batch=8, cap=48, a_sched=1, c_sched=2), 40, 6, "full"),
    "worker_full": ("worker", None, lambda: cav_cfg(cap=512), 400, 4, "digest"),
    "single_time_full": ("single", "time", lambda: light_cfg(cap=512), 300, 4, "digest"),
    "manager_full": ("manager", None, lambda: light_cfg(cap=512), 300, 4, "digest"),
    "triple_full": ("triple", None, lambda: light_cfg(cap=512), 300, 4, "digest"),
}


def opc_cfgs() -> Tuple[dict, dict]:
    """Manager (use_opc=True, one goal per lane: 8) and worker (8 features, 1 goal value, no LSTM)
    configs of the off-policy-correction fixture td3_manager_opc.npz."""
    m = light_cfg(o=(5, 5, 7), T=4, s=8, a=(1, 2, 8), actor_h=(12, 10), critic_h=(14, 12), veh_h=(12, 10),
                  batch=8, cap=48, a_sched=1, c_sched=2)
    m["use_opc"] = True
    return m, cav_cfg(o=8, g=1, hidden=(16, 12), batch=8, cap=48)


def case_seed(case: str, what: str) -> int:
    return _seed(case, what)
