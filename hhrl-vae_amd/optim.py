"""Adam on flat buffers + the StepLR rule, interoperable with torch's
optimizer `state_dict` format (the reference saves `optimizer.state_dict()`,
algorithm.py:132, 333, 335 ...)."""
from __future__ import annotations

import math
from typing import Dict

import torch
from typing import Optional

from . import _lib
from .flat import FlatBuffer, Layout


class FlatAdam:
    """torch.optim.Adam defaults (betas (0.9, 0.999), eps 1e-8, no weight
    decay, no amsgrad) as one fused kernel launch over the flat buffer."""

    beta1, beta2, eps = 0.9, 0.999, 1e-8

    def __init__(self, params: FlatBuffer, lr: float):
        dev = params.data.device
        self.p = params
        self.g = FlatBuffer(params.layout, dev)
        self.m = FlatBuffer(params.layout, dev)
        self.v = FlatBuffer(params.layout, dev)
        self.lr = float(lr)
        self.base_lr = float(lr)
        self.step_count = 0
        self._lib = _lib.load()
        self._hyper_dev = None

    # -- graph-replayable form: scalars in device memory -------------------
    def stage_next_step(self, grad_scale: float = 1.0) -> None:
        """Enqueues the kernel scalars of the NEXT step (count + 1, current lr)
        into device memory on the current stream; `step_dev` launches can then
        be replayed from a captured graph.  Call `commit_step` once the step
        has been enqueued."""
        if self._hyper_dev is None:
            self._hyper_dev = torch.zeros(8, dtype=torch.float32, device=self.p.data.device)
        _lib.check(self._lib.hhrl_adam_stage(self._hyper_dev.data_ptr(), self.lr, self.beta1, self.beta2, self.eps,
                                             self.step_count + 1, float(grad_scale), _lib.stream_ptr()))

    def next_step_scalars(self, grad_scale: float = 1.0):
        """The eight kernel scalars of the NEXT step (count + 1, current lr) as the library computes them
        (bias corrections in double): for callers that carry them to the device inside a transfer of
        their own instead of through `stage_next_step`."""
        t = self.step_count + 1
        bc1 = 1.0 - math.pow(self.beta1, t)
        bc2 = 1.0 - math.pow(self.beta2, t)
        return (self.lr / bc1, self.beta1, self.beta2, 1.0 - self.beta1, 1.0 - self.beta2, math.sqrt(bc2), self.eps,
                float(grad_scale))

    def step_dev(self, hyper_ptr: int = 0, target=None, tau: float = 0.0) -> None:
        """Adam step with the scalars read from device memory (`hyper_ptr`, eight floats; default: the
        buffer filled by `stage_next_step`).  `target` (a FlatBuffer of the same layout): the same launch
        also moves the target network towards the new parameters, target = tau p + (1 - tau) target."""
        hp = hyper_ptr or self._hyper_dev.data_ptr()
        if target is None:
            _lib.check(self._lib.hhrl_adam_step_dev(
                self.p.base_ptr, self.g.base_ptr, self.m.base_ptr, self.v.base_ptr, self.p.layout.size, hp, _lib.stream_ptr()))
        else:
            _lib.check(self._lib.hhrl_adam_polyak_step_dev(
                self.p.base_ptr, self.g.base_ptr, self.m.base_ptr, self.v.base_ptr, self.p.layout.size, hp,
                target.base_ptr, float(tau), _lib.stream_ptr()))

    def commit_step(self) -> None:
        self.step_count += 1

    def zero_grad(self) -> None:
        self.g.data.zero_()

    def step(self, grad_scale: float = 1.0, copy: Optional[tuple] = None) -> None:
        """copy = (src_ptr, dst_ptr): the launch also copies one float (a loss that travelled behind the gradient)."""
        self.step_count += 1
        if copy is None:
            _lib.check(self._lib.hhrl_adam_step(
                self.p.base_ptr, self.g.base_ptr, self.m.base_ptr, self.v.base_ptr, self.p.layout.size,
                self.lr, self.beta1, self.beta2, self.eps, self.step_count, float(grad_scale), _lib.stream_ptr()))
        else:
            _lib.check(self._lib.hhrl_adam_step_copy(
                self.p.base_ptr, self.g.base_ptr, self.m.base_ptr, self.v.base_ptr, self.p.layout.size,
                self.lr, self.beta1, self.beta2, self.eps, self.step_count, float(grad_scale), copy[0], copy[1],
                _lib.stream_ptr()))

    # -- torch-format round trip ------------------------------------------
    def state_dict(self) -> dict:
        names = self.p.layout.names
        state = {}
        if self.step_count > 0:
            for i, n in enumerate(names):
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self.m[n].detach().cpu().clone(),
                            "exp_avg_sq": self.v[n].detach().cpu().clone()}
        group = {"lr": self.lr, "betas": (self.beta1, self.beta2), "eps": self.eps, "weight_decay": 0,
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False,
                 "differentiable": False, "fused": None, "decoupled_weight_decay": False,
                 "initial_lr": self.base_lr, "params": list(range(len(names)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd: dict) -> None:
        names = self.p.layout.names
        group = sd["param_groups"][0]
        if len(group["params"]) != len(names):
            raise ValueError("optimizer state has %d params, network has %d" % (len(group["params"]), len(names)))
        self.lr = float(group["lr"])
        self.base_lr = float(group.get("initial_lr", group["lr"]))
        self.m.data.zero_()
        self.v.data.zero_()
        self.step_count = 0
        for i, n in enumerate(names):
            st = sd["state"].get(i)
            if st is None:
                continue
            self.m[n].copy_(torch.as_tensor(st["exp_avg"], dtype=torch.float32))
            self.v[n].copy_(torch.as_tensor(st["exp_avg_sq"], dtype=torch.float32))
            self.step_count = max(self.step_count, int(float(st["step"])))


class StepLR:
    """torch's StepLR with the reference's gamma = 0.5 (algorithm.py:260-261) in its chainable form:
    every `step_size`-th call multiplies the optimizer's CURRENT lr by gamma.  For a fresh optimizer that
    is base_lr * gamma ** (calls // step_size) bit for bit (halving is exact); after
    `FlatAdam.load_state_dict` it keeps the decayed lr the checkpoint carries and goes on halving it from
    there, as torch does when the reference resumes (algorithm.py:1375-1382: load() restores the optimizer's
    param_groups, the scheduler object keeps counting its own calls).  Computed on the host; the value
    reaches the Adam kernel as a scalar argument."""

    def __init__(self, opt: FlatAdam, step_size: int, gamma: float = 0.5):
        self.opt = opt
        self.step_size = int(step_size)
        self.gamma = float(gamma)
        self.epoch = 0

    def step(self) -> None:
        self.epoch += 1
        if self.epoch % self.step_size == 0:
            self.opt.lr *= self.gamma
