"""TD3 minibatch updates of the hierarchical agents on MI355X.

One engine (`TD3Learner`) covers every variant of the reference
(TD3Single, HATD3Double, HATD3, HATD3Triple, ManagerTD3, WorkerTD3): they
differ only in how many observation streams / actors they have and in the
replay row layout.  The classes at the bottom keep the reference's constructor
signatures, attributes and method names.

A sampled minibatch is uploaded ONCE as a [B, row_width] fp32 block; every
network input is a column slice of that block (pointer + leading dimension),
and every concatenation the reference builds with torch.cat is a column range
of a wider activation buffer that producers write into directly.
"""
from __future__ import annotations

import copy
import time
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib, ops
from ._lib import ACT_NONE, ACT_RELU, ACT_TANH, check
from .flat import FlatBuffer, Layout
from .ops import BiLstm, Lin, P
from .optim import FlatAdam, StepLR
from .spec import Family, actor_shapes, critic_shapes, family_from_cfg
from .vae import _default_init


def _lstm_default_init(layout: Layout, gen: torch.Generator) -> Dict[str, torch.Tensor]:
    """nn.Linear default for dense layers, nn.LSTM default U(-1/sqrt(H), 1/sqrt(H)) for LSTM tensors."""
    out = _default_init(layout, gen)
    for name, shp in layout.shapes.items():
        if "lstm_layer." in name:
            hid = shp[0] // 4
            out[name] = (torch.rand(shp, generator=gen) * 2 - 1) / (hid ** 0.5)
    return out


import os as _os

_DP_DEBUG = _os.environ.get("HHRL_DP_DEBUG") == "1"       # diagnostic: synchronise and report after every stretch


# ------------------------------------------------------------------ networks
class _ActorWork:
    def __init__(self, net: "ActorNet", B: int, dev):
        e = lambda *s: ops.dev_empty(*s, device=dev)
        h0, h1 = net.h
        self.xin = e(B, net.d_in) if net.needs_xin else None
        self.h1, self.h2 = e(B, h0), e(B, h1)
        self.d_h2, self.d_h1 = e(B, h1), e(B, h0)
        self.d_xin = e(B, net.d_in) if net.lstm is not None else None
        self.lstm_ws = net.lstm.workspace(B, dev) if net.lstm is not None else None


class ActorNet:
    """Actor (algorithm.py:159-180) / WorkerActor (:1187-1211)."""

    def __init__(self, fam: Family, k: int, device, trainable: bool, lr: float = 0.0):
        self.fam, self.k, self.dev = fam, k, device
        self.stream = fam.streams[k]
        self.layout = Layout(actor_shapes(fam, k))
        self.params = FlatBuffer(self.layout, device)
        self.opt = FlatAdam(self.params, lr) if trainable else None
        g = self.opt.g if trainable else None
        s = self.stream
        self.lstm = BiLstm(self.params, g, "lstm.", s.o, s.T, s.s) if s.T > 1 else None
        self.d_in = s.feat + fam.goal_dim
        self.needs_xin = self.lstm is not None or fam.goal_dim > 0
        self.h = tuple(fam.actor_hidden)
        self.a = fam.a_dims[k]
        self.l1 = Lin(self.params, g, "l1", ACT_RELU)
        self.l2 = Lin(self.params, g, "l2", ACT_RELU)
        self.l3 = Lin(self.params, g, "l3", ACT_TANH)          # max_action == 1
        # 128-wide networks without an LSTM run as one fused launch per direction
        self.mlp3 = ops.Mlp3(self.l1, self.l2, self.l3) if self.lstm is None and ops.Mlp3.supported(self.l1, self.l2, self.l3) else None
        self._work: Dict[int, _ActorWork] = {}

    def work(self, B: int) -> _ActorWork:
        w = self._work.get(B)
        if w is None:
            w = self._work[B] = _ActorWork(self, B, self.dev)
        return w

    def lstm_item(self, B: int, obs: int, ld_obs: int, no_backward: bool = False):
        """This actor's LSTM evaluation as an entry of a grouped launch (ops.lstm_forward_group)."""
        w = self.work(B)
        return (self.lstm, obs, ld_obs, P(w.xin), self.d_in, w.lstm_ws, no_backward)

    def lstm_bwd_item(self, B: int):
        w = self.work(B)
        return (self.lstm, P(w.d_xin), self.d_in, w.lstm_ws)

    def stage_inputs(self, cq: "ops.CopyQueue", B: int, obs: int, ld_obs: int, goal: Optional[int], ld_goal: int) -> None:
        """Queues the copies that build the MLP input from replay-row columns (raw observation when
        there is no LSTM, goal)."""
        w = self.work(B)
        s = self.stream
        if self.needs_xin:
            if self.lstm is None:
                cq.add(obs, ld_obs, P(w.xin), self.d_in, s.o)
            if self.fam.goal_dim:
                cq.add(goal, ld_goal, P(w.xin, s.feat), self.d_in, self.fam.goal_dim)

    def bind_input(self, B: int, obs: int, ld_obs: int) -> None:
        w = self.work(B)
        w.x, w.ldx = (P(w.xin), self.d_in) if self.needs_xin else (obs, ld_obs)

    def level_item(self, level: int, B: int, out: int, ld_out: int):
        """Layer `level` (1..3) of the MLP as an entry of ops.linear_forward_group (after bind_input)."""
        w = self.work(B)
        if level == 1:
            return (self.l1, w.x, w.ldx, P(w.h1), self.h[0])
        if level == 2:
            return (self.l2, P(w.h1), self.h[0], P(w.h2), self.h[1])
        return (self.l3, P(w.h2), self.h[1], out, ld_out)

    def fused_desc(self, d, B: int, out: int, ld_out: int) -> None:
        """Fills one hhrl_mlp3_desc with this actor's forward pass (after bind_input)."""
        w = self.work(B)
        self.mlp3.desc(d, w.x, w.ldx, out, ld_out, P(w.h1), P(w.h2))

    def forward(self, B: int, obs: int, ld_obs: int, goal: Optional[int], ld_goal: int, out: int, ld_out: int,
                lstm_done: bool = False, inputs_done: bool = False, no_backward: bool = False) -> None:
        """lstm_done: the LSTM feature was already produced by a grouped launch built from lstm_item();
        inputs_done: stage_inputs() was already flushed for these rows; no_backward: backward() will not
        follow on this evaluation (the LSTM then saves none of its steps)."""
        w = self.work(B)
        if self.lstm is not None and not lstm_done:
            if no_backward:
                ops.lstm_forward_group([self.lstm_item(B, obs, ld_obs, True)], B)
            else:
                self.lstm.fwd(obs, ld_obs, P(w.xin), self.d_in, w.lstm_ws, B)
        if not inputs_done:
            cq = ops.CopyQueue(B)
            self.stage_inputs(cq, B, obs, ld_obs, goal, ld_goal)
            if cq.items:
                cq.flush()
        self.bind_input(B, obs, ld_obs)
        x, ldx = w.x, w.ldx
        if self.mlp3 is not None:
            d = ops.mlp3_descs(1)
            self.fused_desc(d[0], B, out, ld_out)
            ops.mlp3_forward_group(d, B)
            return
        self.l1.fwd(x, ldx, P(w.h1), self.h[0], B)
        self.l2.fwd(P(w.h1), self.h[0], P(w.h2), self.h[1], B)
        self.l3.fwd(P(w.h2), self.h[1], out, ld_out, B)

    def backward(self, B: int, out: int, ld_out: int, d_out: int, ld_dout: int, lstm: bool = True, ride=None) -> bool:
        """Gradients of the most recent forward(B, ...) into this actor's flat
        gradient buffer (added; the caller zeroes it).  lstm=False leaves the
        BPTT to the caller (lstm_bwd_item).  ride = (hyper_ptr, target FlatBuffer or None, tau): the optimizer step
        may follow inside the same call (fused networks only); returns True when it did."""
        w = self.work(B)
        if self.mlp3 is not None:
            d = ops.mlp3_descs(1)
            self.fused_desc(d[0], B, out, ld_out)
            self.mlp3.grads(d[0], d_out, ld_dout, 0.0, True, None, 0, self.mlp3.slab(B, self.dev))
            ops.mlp3_backward_group(d, B, (self.opt,) + tuple(ride) if ride is not None else None)
            return ride is not None
        self.l3.bwd(P(w.h2), self.h[1], out, ld_out, d_out, ld_dout, P(w.d_h2), self.h[1], B)
        self.l2.bwd(P(w.h1), self.h[0], P(w.h2), self.h[1], P(w.d_h2), self.h[1], P(w.d_h1), self.h[0], B)
        if self.lstm is not None:
            self.l1.bwd(w.x, w.ldx, P(w.h1), self.h[0], P(w.d_h1), self.h[0], P(w.d_xin), self.d_in, B)
            if lstm:
                self.lstm.bwd(P(w.d_xin), self.d_in, w.lstm_ws, B)
        else:
            self.l1.bwd(w.x, w.ldx, P(w.h1), self.h[0], P(w.d_h1), self.h[0], None, 0, B)
        return False


class _CriticWork:
    def __init__(self, net: "CriticNet", B: int, dev):
        e = lambda *s: ops.dev_empty(*s, device=dev)
        h0, h1 = net.h
        self.xu = e(B, net.D)
        self.xu_pi = e(B, net.D)       # head input of the policy step (state, policy action) when the actor runs ahead of the critic step
        self.a1, self.a2, self.q1 = e(B, h0), e(B, h1), e(B, 1)
        self.b1, self.b2, self.q2 = e(B, h0), e(B, h1), e(B, 1)
        self.d_q1, self.d_q2 = e(B, 1), e(B, 1)
        self.d_qc: Dict[float, torch.Tensor] = {}           # dL/dQ1 of a policy step (one constant for every row), by constant
        self.d_a2, self.d_a1 = e(B, h1), e(B, h0)
        self.d_b2, self.d_b1 = e(B, h1), e(B, h0)
        self.d_xu = e(B, net.D)
        self.lstm_ws = [l.workspace(B, dev) if l is not None else None for l in net.lstms]


class CriticNet:
    """CriticSingle / CriticDouble / CriticTriple / ManagerCritic / WorkerCritic
    (algorithm.py:188-232, 352-405, 614-676, 867-911, 1214-1258): per-stream
    BiLSTM (when T > 1), concatenation with [goal and] the joint action, two
    independent Q heads."""

    def __init__(self, fam: Family, device, trainable: bool, lr: float = 0.0):
        self.fam, self.dev = fam, device
        self.layout = Layout(critic_shapes(fam))
        self.params = FlatBuffer(self.layout, device)
        self.opt = FlatAdam(self.params, lr) if trainable else None
        g = self.opt.g if trainable else None
        self.lstms = [BiLstm(self.params, g, pre, s.o, s.T, s.s) if s.T > 1 else None
                      for s, pre in zip(fam.streams, fam.critic_lstm_prefixes())]
        self.D = fam.critic_in
        self.h = tuple(fam.critic_hidden)
        self.feat_cols: List[int] = []
        col = 0
        for s in fam.streams:
            self.feat_cols.append(col)
            col += s.feat
        self.goal_col = col
        col += fam.goal_dim
        self.act_col = col
        self.act_cols = [col + sum(fam.a_dims[:k]) for k in range(len(fam.a_dims))]
        L = lambda n, a: Lin(self.params, g, n, a)
        self.l1, self.l2, self.l3 = L("l1", ACT_RELU), L("l2", ACT_RELU), L("l3", ACT_NONE)
        self.l4, self.l5, self.l6 = L("l4", ACT_RELU), L("l5", ACT_RELU), L("l6", ACT_NONE)
        # 128-wide heads without LSTM streams run as one fused launch per direction (both heads side by side)
        self.mlp3 = None
        if all(l is None for l in self.lstms) and ops.Mlp3.supported(self.l1, self.l2, self.l3):
            self.mlp3 = (ops.Mlp3(self.l1, self.l2, self.l3), ops.Mlp3(self.l4, self.l5, self.l6))
        self._work: Dict[int, _CriticWork] = {}

    def work(self, B: int) -> _CriticWork:
        w = self._work.get(B)
        if w is None:
            w = self._work[B] = _CriticWork(self, B, self.dev)
        return w

    def feature(self, k: int, B: int, o_ptr: int, o_ld: int) -> None:
        """Stream k's part of the head input: its LSTM feature or its raw observation."""
        w = self.work(B)
        s, lstm, col = self.fam.streams[k], self.lstms[k], self.feat_cols[k]
        if lstm is not None:
            lstm.fwd(o_ptr, o_ld, P(w.xu, col), self.D, w.lstm_ws[k], B)
        else:
            ops.copy2d(o_ptr, o_ld, P(w.xu, col), self.D, B, s.o)

    def feature_item(self, k: int, B: int, o_ptr: int, o_ld: int, no_backward: bool = False):
        """Stream k's LSTM evaluation as an entry of a grouped launch (streams without an LSTM: None)."""
        if self.lstms[k] is None:
            return None
        w = self.work(B)
        return (self.lstms[k], o_ptr, o_ld, P(w.xu, self.feat_cols[k]), self.D, w.lstm_ws[k], no_backward)

    def lstm_bwd_item(self, k: int, B: int):
        if self.lstms[k] is None:
            return None
        w = self.work(B)
        return (self.lstms[k], P(w.d_xu, self.feat_cols[k]), self.D, w.lstm_ws[k])

    def stage_inputs(self, cq: "ops.CopyQueue", B: int, obs: Sequence[Tuple[int, int]], goal: Optional[int], ld_goal: int,
                     actions: Optional[int] = None, ld_act: int = 0, policy_input: bool = False) -> None:
        """Queues the copies of replay-row columns into the head input (policy_input: into the policy step's
        own input buffer): raw observations of streams without an LSTM, the goal, and (if given) the stored
        joint action."""
        w = self.work(B)
        xu = w.xu_pi if policy_input else w.xu
        for k, (o_ptr, o_ld) in enumerate(obs):
            if self.lstms[k] is None:
                cq.add(o_ptr, o_ld, P(xu, self.feat_cols[k]), self.D, self.fam.streams[k].o)
        if self.fam.goal_dim:
            cq.add(goal, ld_goal, P(xu, self.goal_col), self.D, self.fam.goal_dim)
        if actions is not None:
            cq.add(actions, ld_act, P(xu, self.act_col), self.D, self.fam.act_dim)

    def set_goal(self, B: int, goal: Optional[int], ld_goal: int) -> None:
        if self.fam.goal_dim:
            w = self.work(B)
            ops.copy2d(goal, ld_goal, P(w.xu, self.goal_col), self.D, B, self.fam.goal_dim)

    def features(self, B: int, obs: Sequence[Tuple[int, int]], goal: Optional[int], ld_goal: int) -> None:
        """State part of the head input: per-stream features, then the goal."""
        for k, (o_ptr, o_ld) in enumerate(obs):
            self.feature(k, B, o_ptr, o_ld)
        self.set_goal(B, goal, ld_goal)

    def set_actions(self, B: int, src: int, ld_src: int) -> None:
        """Joint action block copied from `src` (replay rows keep a1|a2|a3 adjacent)."""
        w = self.work(B)
        ops.copy2d(src, ld_src, P(w.xu, self.act_col), self.D, B, self.fam.act_dim)

    def level_items(self, level: int, B: int):
        """Layer `level` (1..3) of both heads on w.xu as entries of ops.linear_forward_group."""
        w = self.work(B)
        h0, h1 = self.h
        if level == 1:
            return [(self.l1, P(w.xu), self.D, P(w.a1), h0), (self.l4, P(w.xu), self.D, P(w.b1), h0)]
        if level == 2:
            return [(self.l2, P(w.a1), h0, P(w.a2), h1), (self.l5, P(w.b1), h0, P(w.b2), h1)]
        return [(self.l3, P(w.a2), h1, P(w.q1), 1), (self.l6, P(w.b2), h1, P(w.q2), 1)]

    def head_descs(self, d, start: int, B: int, q1_only: bool = False, policy_input: bool = False) -> None:
        """Fills d[start] (and d[start + 1]) with the forward pass of head 1 (and head 2) on w.xu (w.xu_pi)."""
        w = self.work(B)
        xu = w.xu_pi if policy_input else w.xu
        self.mlp3[0].desc(d[start], P(xu), self.D, P(w.q1), 1, P(w.a1), P(w.a2))
        if not q1_only:
            self.mlp3[1].desc(d[start + 1], P(xu), self.D, P(w.q2), 1, P(w.b1), P(w.b2))

    def heads(self, B: int, q1_only: bool = False, policy_input: bool = False, hidden_only: bool = False) -> None:
        """hidden_only (with q1_only, layer-by-layer heads): the two hidden layers without the output layer -- all the
        policy step needs, whose gradient d(-mean Q1)/d(hidden 2) is a row of constants (backward(dq_const=...))."""
        w = self.work(B)
        h0, h1 = self.h
        if self.mlp3 is not None:
            d = ops.mlp3_descs(1 if q1_only else 2)
            self.head_descs(d, 0, B, q1_only, policy_input)
            ops.mlp3_forward_group(d, B)
            return
        assert not policy_input, "the policy step's own input buffer is used with the fused three-layer heads only"
        if q1_only:
            self.l1.fwd(P(w.xu), self.D, P(w.a1), h0, B)
            self.l2.fwd(P(w.a1), h0, P(w.a2), h1, B)
            if not hidden_only:
                self.l3.fwd(P(w.a2), h1, P(w.q1), 1, B)
            return
        # the two heads are independent: layer k of both in one launch
        for level in (1, 2, 3):
            ops.linear_forward_group(self.level_items(level, B), B)

    def q1_input_gradient(self, B: int, dq_const: float, policy_input: bool = False) -> bool:
        """Head 1 forward and the input gradient for dL/dQ1 = dq_const on every row (w.d_xu) as ONE launch, where
        the fused three-layer kernels serve the heads; False where they do not (the caller runs heads + backward)."""
        if self.mlp3 is None:
            return False
        w = self.work(B)
        d = ops.mlp3_descs(1)
        self.mlp3[0].desc(d[0], P(w.xu_pi if policy_input else w.xu), self.D, P(w.q1), 1, P(w.a1), P(w.a2))
        self.mlp3[0].grads(d[0], None, 1, dq_const, False, P(w.d_xu), self.D)
        d[0].forward_dx = 1
        ops.mlp3_forward_group(d, B)
        return True

    def fused_target(self) -> bool:
        """Whether backward(both_heads=True, param_grads=True, target=...) forms the critic loss inside its launch."""
        return self.mlp3 is not None and not any(l is not None for l in self.lstms)

    def lstm_backward(self, k: int, B: int) -> None:
        """BPTT of stream k's LSTM from its columns of w.d_xu (independent of the other streams)."""
        w = self.work(B)
        if self.lstms[k] is not None:
            self.lstms[k].bwd(P(w.d_xu, self.feat_cols[k]), self.D, w.lstm_ws[k], B)

    def backward(self, B: int, both_heads: bool, param_grads: bool, lstms: bool = True,
                 dq_const: Optional[float] = None, policy_input: bool = False, dx_cols: Optional[Tuple[int, int]] = None,
                 ride=None, target=None) -> bool:
        """Back-propagates w.d_q1 (and w.d_q2) to w.d_xu; with param_grads also
        into the critic's gradient buffer (its LSTMs too unless lstms=False:
        the caller then runs lstm_backward per stream).  dq_const: dL/dQ1 is
        this constant for every row (the gradient of -mean Q1) instead of w.d_q1.  dx_cols = (first column, count):
        only those columns of w.d_xu are wanted (one actor's action columns on a policy step); honoured where it
        saves work, else the whole of w.d_xu is written.  ride = (hyper_ptr, target FlatBuffer or None, tau): the
        optimizer step may follow inside the same call (both fused heads with their parameter gradients: all of the
        optimizer's parameters); returns True when it did.  target = (q1t, q2t, reward, ld_reward, gamma, inv_n, loss,
        y): with both fused heads the launch forms the TD3 target, both losses (added to loss[0..1], zeroed by the
        caller) and dL/dQ itself instead of reading w.d_q1 / w.d_q2 (see fused_target())."""
        w = self.work(B)
        h0, h1 = self.h
        has_lstm = any(l is not None for l in self.lstms)
        need_dxu = (not param_grads) or has_lstm
        dxu = P(w.d_xu) if need_dxu else None
        if self.mlp3 is not None and not (both_heads and need_dxu):
            d = ops.mlp3_descs(2 if both_heads else 1)
            self.mlp3[0].desc(d[0], P(w.xu_pi if policy_input else w.xu), self.D, P(w.q1), 1, P(w.a1), P(w.a2))
            self.mlp3[0].grads(d[0], None if dq_const is not None else P(w.d_q1), 1, dq_const or 0.0, param_grads, dxu, self.D,
                               self.mlp3[0].slab(B, self.dev) if param_grads else None)
            if both_heads:
                self.mlp3[1].desc(d[1], P(w.xu), self.D, P(w.q2), 1, P(w.b1), P(w.b2))
                self.mlp3[1].grads(d[1], P(w.d_q2), 1, 0.0, param_grads, None, 0,
                                   self.mlp3[1].slab(B, self.dev) if param_grads else None)
            if target is not None:
                assert both_heads and dq_const is None
                q1t, q2t, r, ldr, gamma, inv_n, loss, y = target
                for k in (0, 1):
                    d[k].dy, d[k].mse_q1t, d[k].mse_q2t, d[k].mse_r, d[k].mse_ldr = None, q1t, q2t, r, ldr
                    d[k].mse_gamma, d[k].mse_inv_n, d[k].mse_loss, d[k].mse_target = gamma, inv_n, loss + 4 * k, y
            rides = ride is not None and both_heads and param_grads
            ops.mlp3_backward_group(d, B, (self.opt,) + tuple(ride) if rides else None)
            return rides
        assert target is None, "the TD3 target is formed by the fused three-layer heads only (fused_target())"
        d_q1 = P(w.d_q1)
        assert not policy_input, "the policy step's own input buffer is used with the fused three-layer heads only"
        if dq_const is not None and not both_heads and not param_grads and self.l3.N == 1:
            # dL/dQ1 is the same constant for every row, so dL/d(hidden 2) is dq_const times the output layer's weight row
            # for every sample: the output layer takes no launch, its row is the operand of the second layer's product
            ops.linear_backward_row(self.l2, P(w.a2), h1, self.l3.w, dq_const, P(w.d_a1), h0, B)
            if dx_cols is not None:
                ops.linear_backward_cols(self.l1, P(w.a1), h0, P(w.d_a1), h0, dxu, self.D, dx_cols[0], dx_cols[1], B)
            else:
                self.l1.bwd(P(w.xu), self.D, P(w.a1), h0, P(w.d_a1), h0, dxu, self.D, B, param_grads=False)
            return
        if dq_const is not None:
            # The constant gradient lives in a buffer of its own, one per constant (the divisor is the global batch,
            # which a caller may vary at one B): filled once, on the first -- eager -- run of a launch chain, so a
            # captured chain carries no fill and still reads ITS constant whatever other chains ran in between.
            buf = w.d_qc.get(dq_const)
            if buf is None:
                if len(w.d_qc) >= 64:
                    raise RuntimeError("more than 64 different global batch sizes at one shard size")
                buf = w.d_qc[dq_const] = ops.dev_empty(B, 1, device=self.dev)
                ops.fill(P(buf), B, dq_const)
            d_q1 = P(buf)
        if both_heads:
            # layer k of both heads per stage; the two first layers add up in d_xu (head 2 accumulates)
            ops.linear_backward_group([(self.l3, P(w.a2), h1, None, 0, P(w.d_q1), 1, P(w.d_a2), h1, 0),
                                       (self.l6, P(w.b2), h1, None, 0, P(w.d_q2), 1, P(w.d_b2), h1, 0)], B, param_grads)
            ops.linear_backward_group([(self.l2, P(w.a1), h0, P(w.a2), h1, P(w.d_a2), h1, P(w.d_a1), h0, 0),
                                       (self.l5, P(w.b1), h0, P(w.b2), h1, P(w.d_b2), h1, P(w.d_b1), h0, 0)], B, param_grads)
            ops.linear_backward_group([(self.l1, P(w.xu), self.D, P(w.a1), h0, P(w.d_a1), h0, dxu, self.D, 0),
                                       (self.l4, P(w.xu), self.D, P(w.b1), h0, P(w.d_b1), h0, dxu, self.D, 1)], B, param_grads)
        else:
            self.l3.bwd(P(w.a2), h1, None, 0, d_q1, 1, P(w.d_a2), h1, B, param_grads=param_grads)
            self.l2.bwd(P(w.a1), h0, P(w.a2), h1, P(w.d_a2), h1, P(w.d_a1), h0, B, param_grads=param_grads)
            self.l1.bwd(P(w.xu), self.D, P(w.a1), h0, P(w.d_a1), h0, dxu, self.D, B, param_grads=param_grads)
        if param_grads and lstms:
            for k in range(len(self.lstms)):
                self.lstm_backward(k, B)


# ------------------------------------------------------------------ replay
class DictReplay:
    """ManagerTD3.ReplayBuffer (algorithm.py:916-951): fp32 arrays by field."""

    def __init__(self, capacity: int, obs_dim: int, act_dim: int, seq_len: int, use_opc: bool):
        self.capacity = capacity
        self.buffer = {
            "obs": np.zeros((capacity, obs_dim), dtype=np.float32),
            "next_obs": np.zeros((capacity, obs_dim), dtype=np.float32),
            "action": np.zeros((capacity, act_dim), dtype=np.float32),
            "reward": np.zeros((capacity, 1), dtype=np.float32),
            "worker_obs": np.zeros((capacity, seq_len, 8, 8), dtype=np.float32),
            "worker_act": np.zeros((capacity, seq_len, 8, 1), dtype=np.float32),
        }
        self.pointer = 0
        self.useOPC = use_opc

    def store_transition(self, *args):
        o, a, r, o_, wo, wa = args
        i = self.pointer % self.capacity
        self.buffer["obs"][i] = o
        self.buffer["action"][i] = a
        self.buffer["reward"][i] = r
        self.buffer["next_obs"][i] = o_
        if self.useOPC:
            self.buffer["worker_obs"][i] = np.array(wo)
            self.buffer["worker_act"][i] = np.array(wa)
        self.pointer += 1

    def __getitem__(self, item):
        return self.buffer[item]


# ------------------------------------------------------------------ learner
class TD3Learner:
    """The update rule shared by every variant: WorkerTD3.learn
    (algorithm.py:1316-1366), TD3Single.learn (:283-328), ManagerTD3.learn
    (:1109-1168), HATD3Double.learn (:479-542), HATD3Triple.learn (:768-838).
    """

    AUTO_DEVICE_REPLAY_FLOATS = 32768

    def __init__(self, fam: Family, device: Optional[str] = None, seed: Optional[int] = None, group=None,
                 use_graph=True, device_replay="auto"):
        if not torch.cuda.is_available():
            raise RuntimeError("hhrl_vae_amd needs an AMD GPU (no CPU fallback)")
        self._lib = _lib.load()
        self.fam = fam
        self.device = torch.device(device or "cuda")
        dev = self.device
        gen = torch.Generator().manual_seed(3407 if seed is None else seed)
        self.actor_names = fam.actor_names()
        self.actors = [ActorNet(fam, k, dev, True, fam.actor_lr) for k in range(len(fam.a_dims))]
        self.actor_targets = [ActorNet(fam, k, dev, False) for k in range(len(fam.a_dims))]
        for a, t in zip(self.actors, self.actor_targets):
            a.params.load(_lstm_default_init(a.layout, gen))
            t.params.copy_from(a.params)                                # algorithm.py:258
        self.critic = CriticNet(fam, dev, True, fam.critic_lr)
        self.critic_target = CriticNet(fam, dev, False)
        self.critic.params.load(_lstm_default_init(self.critic.layout, gen))
        self.critic_target.params.copy_from(self.critic.params)        # :265
        self.actor_scheds = [StepLR(a.opt, fam.actor_sched) for a in self.actors]
        self.critic_sched = StepLR(self.critic.opt, fam.critic_sched)
        self.policy_freq = fam.policy_freq
        self.total_it = 0
        self.batch_size = fam.batch_size
        self.gamma, self.tau = fam.gamma, fam.tau
        self.memory_capacity = fam.memory_capacity
        self.learn_begin = fam.memory_capacity * fam.learn_start_ratio
        self.group = group
        self.world_size, self.rank = 1, 0
        self._group_backend = None
        if group is not None:
            import torch.distributed as dist
            self.world_size, self.rank = dist.get_world_size(group), dist.get_rank(group)
            self._group_backend = str(dist.get_backend(group))
        # smoothing noise (algorithm.py:295, 1330).  Data-parallel ranks share `seed` (identical initial weights)
        # but hold different shards of the global minibatch: each rank's noise stream is offset by its rank, so
        # that the global batch carries independent draws like the single-device update of the same rows would
        self._np_rng = np.random.default_rng((3407 if seed is None else seed) + 1000003 * self.rank)
        self._noise_seed = (3407 if seed is None else int(seed)) + 1000003 * self.rank      # device draws: the stream's key
        self._noise_drawn = 0                                                               # ... and how far it has been used
        self._loss = ops.dev_zeros(2, device=dev)
        self._statics: Dict[int, "TD3Learner._Static"] = {}
        self._policy_io: Dict[Tuple[int, int], dict] = {}
        self._graphs: Dict[tuple, "torch.cuda.CUDAGraph"] = {}
        # data-parallel: a launch chain is captured as one graph per stretch between two all-reduces
        self._seg_graphs: Dict[tuple, list] = {}
        self._sync_hook = None
        # graph_collectives (RCCL groups only; HHRL_DP_GRAPH_COLLECTIVE=1 or the attribute set before the first
        # update): the gradient all-reduces are captured INSIDE one graph per update instead of running between
        # segment graphs.  Off by default: a collective between two replays is the form every c10d backend supports.
        self.graph_collectives = _os.environ.get("HHRL_DP_GRAPH_COLLECTIVE") == "1" and self._group_backend == "nccl"
        self._seen: Dict[tuple, int] = {}
        self._slot_graphs: Dict[tuple, int] = {}        # launch chain -> graphs captured so far, one per staging block
        if use_graph not in (True, False, "auto"):
            raise ValueError("use_graph must be True, False or 'auto'")
        self.use_graph = use_graph
        # device_replay: the flat replay buffer is mirrored in device memory (new rows are uploaded before an
        # update) and a minibatch is gathered there from the sampled row numbers, instead of being gathered
        # on the host and uploaded whole.  Classes with a flat `memory` array only.  Data-parallel ranks each
        # mirror their own replay buffer (a rank samples its shard of the global minibatch from its own rows).
        # "auto" (default): on when a minibatch is at least AUTO_DEVICE_REPLAY_FLOATS wide x long, where gathering
        # it on the host and uploading it costs more than the extra gather launch (HATD3Triple at batch 64:
        # 0.715 -> 0.682 ms; WorkerTD3 at batch 4096: 0.181 -> 0.136 ms; WorkerTD3 at batch 64: 0.090 -> 0.099 ms)
        if device_replay == "auto":
            device_replay = fam.batch_size * fam.row_width >= self.AUTO_DEVICE_REPLAY_FLOATS
        self.device_replay = bool(device_replay)
        self._tune: Dict[tuple, dict] = {}
        # column offsets inside a replay row
        f = fam
        pos = 0
        self.col_obs = []
        for s in f.streams:
            self.col_obs.append(pos); pos += s.width
        self.col_act = pos; pos += f.act_dim
        self.col_goal = pos; pos += f.goal_dim
        self.col_next_goal = pos; pos += f.goal_dim
        self.col_reward = pos; pos += 1
        self.col_next_obs = []
        for s in f.streams:
            self.col_next_obs.append(pos); pos += s.width
        assert pos == f.row_width
        self.last_critic_loss: Optional[torch.Tensor] = None

    # ---- batches ----------------------------------------------------------
    def _sample_rows(self, indices: np.ndarray) -> np.ndarray:
        raise NotImplementedError

    def _n_stored(self) -> int:
        raise NotImplementedError

    # ---- the update -------------------------------------------------------
    FETCH_MAX_FLOATS = 1 << 21          # largest staging block an update fetches itself (first launch of its chain)

    class _Static:
        """Per-batch-size device buffers with fixed addresses (a captured graph replays on them).

        Everything an update needs from the host -- the sampled rows or their row numbers, the smoothing noise
        (or the seed and offset of its stream), the scalars of the Adam steps -- sits in ONE device block, refreshed
        once per update from a pinned staging block of the update's launch chain (`_Ring`): by the update's first
        launch where the block is small, by one transfer in front of the update otherwise."""

        def __init__(self, fam: Family, B: int, dev, n_opt: int):
            a64 = lambda n: (n + 63) & ~63
            W = fam.row_width
            self.B, self.a_dims = B, tuple(fam.a_dims)
            self.n_rows = B * W
            off = a64(self.n_rows)
            self.noise_off = []
            for a in fam.a_dims:
                self.noise_off.append(off)
                off += a64(B * a)
            self.hyper_off = []
            for _ in range(n_opt):
                self.hyper_off.append(off)
                off += 64
            self.idx_off = off                 # B int32 row numbers of a device-resident replay buffer
            off += a64(B)
            self.total = off
            self.d_in = ops.dev_zeros(off, device=dev)
            self.rows = self.d_in[:self.n_rows].view(B, W)
            self.noise = [self.d_in[o:o + B * a].view(B, a) for o, a in zip(self.noise_off, fam.a_dims)]
            self.hyper = [self.d_in[o:o + 8] for o in self.hyper_off]
            self.idx = self.d_in[self.idx_off:self.idx_off + B].view(torch.int32)
            self.rings: Dict[tuple, "TD3Learner._Ring"] = {}
            self.upload = None                 # (pinned source, device destination, floats) when the update fetches its block itself
            self.pi = [ops.dev_empty(B, a, device=dev) for a in fam.a_dims]
            self.y = ops.dev_empty(B, 1, device=dev)

        def ring(self, key: tuple, row_width: int = 0) -> "TD3Learner._Ring":
            """row_width > 0: the blocks have room for rows on their way into a device-resident replay buffer."""
            r = self.rings.get(key)
            if r is None:
                r = self.rings[key] = TD3Learner._Ring(self, row_width)
            return r

    class _Ring:
        """The pinned staging blocks of ONE launch chain: the host fills one while the device reads the other.  A chain
        has blocks of its own because its captured graph names them: the graph's first node fetches the block."""
        SLOTS = 2
        NEW_ROWS_MAX = 16         # replay rows stored since the last update that may ride in the block (<= 63)

        def __init__(self, st: "TD3Learner._Static", row_width: int = 0):
            B = st.B
            # behind the part that mirrors the device block: {count, positions} and the rows themselves
            self.new_off = st.total
            extra = 64 + self.NEW_ROWS_MAX * row_width if row_width else 0
            self.h_in = [torch.zeros(st.total + extra, dtype=torch.float32).pin_memory() for _ in range(self.SLOTS)]
            self.h_np = [t.numpy() for t in self.h_in]
            self.h_ptr = [t.data_ptr() for t in self.h_in]
            self.h_idx = [t[st.idx_off:st.idx_off + B].view(torch.int32).numpy() for t in self.h_in]
            self.h_noise = [[h[o:o + B * a].reshape(B, a) for o, a in zip(st.noise_off, st.a_dims)] for h in self.h_np]
            self.h_hyper = [[h[o:o + 8] for o in st.hyper_off] for h in self.h_np]
            # seed and stream offset of an update that draws its smoothing noise on the device: two 64-bit integers in
            # the spare part of the first optimizer's scalars
            self.rng_off = st.hyper_off[0] + 16
            self.h_rng = [h[self.rng_off:self.rng_off + 4].view(np.uint64) for h in self.h_np]
            if row_width:
                self.h_new_hdr = [h[self.new_off:self.new_off + 64].view(np.int32) for h in self.h_np]
                self.h_new_rows = [h[self.new_off + 64:].reshape(self.NEW_ROWS_MAX, row_width) for h in self.h_np]
            self.zero_copy = all(_lib.zero_copy_ok(t, any_size=True) for t in self.h_in)      # kernels may read the blocks in place
            self.events: List[Optional["torch.cuda.Event"]] = [None] * self.SLOTS
            self.turn = 0

        def next_slot(self) -> int:
            slot = self.turn % self.SLOTS
            self.turn += 1
            if self.events[slot] is not None:
                self.events[slot].synchronize()
            return slot

        def release_after(self, slot: int, stream) -> None:
            """The slot may be refilled once the work enqueued on `stream` so far has run."""
            ev = self.events[slot]
            if ev is None:
                ev = self.events[slot] = torch.cuda.Event()
            ev.record(stream)

    def learn_rows(self, rows, noise: Optional[Sequence] = None, order: Optional[Sequence[int]] = None,
                   _indices=None, global_batch: Optional[int] = None, _new_rows=None) -> None:
        """One TD3 update on a [B, row_width] block of replay rows (numpy array, host or device tensor).
        _indices (device replay, rows=None): the B row numbers to gather from the device mirror instead;
        _new_rows: positions of rows stored since the last update that are not in the mirror yet (none of them among
        _indices): they travel with this update where its first launch fetches the staging block, else by _sync_replay().

        The kernel sequence of an update depends only on (B, policy step?, actor order); after two
        eager runs of a given combination it is captured into a HIP graph and replayed.  What changes
        from update to update (rows, noise, the scalars of the Adam steps) sits in one pinned block that
        the update's first launch fetches (small blocks; one graph per block) or one transfer brings over
        before the replay.

        noise: one [B, a] array per actor, the standard normals of the target-policy smoothing
        (algorithm.py:790, 1333).  None: drawn here -- on the device, in the update's first launch, by a
        single-device learner with graphs (the stream of hhrl_randn_fill for this learner's seed), else
        from this learner's numpy generator.

        global_batch (data-parallel): rows of the global minibatch over all ranks, the divisor of the batch
        means (default: this rank's rows x world size, i.e. equal shards)."""
        fam = self.fam
        W = fam.row_width
        gather = _indices is not None
        if gather:
            _indices = np.asarray(_indices)
            B = int(_indices.shape[0])
            if _indices.ndim != 1 or B == 0:
                raise ValueError("indices must be a non-empty vector")
        else:
            B = int(rows.shape[0])
            if rows.ndim != 2 or rows.shape[1] != W or B == 0:
                raise ValueError("rows must be [B, %d], got %s" % (W, tuple(rows.shape)))
        st = self._statics.get(B)
        if st is None:
            st = self._statics[B] = TD3Learner._Static(fam, B, self.device, 1 + len(self.actors))
        n_act = len(self.actors)
        self.total_it += 1
        policy = self.total_it % self.policy_freq == 0
        if policy:
            if order is None:
                order = np.random.permutation(n_act) if n_act > 1 else [0]
                if self.group is not None and n_act > 1:          # every rank walks the actors in rank 0's order
                    from . import dist as hdist
                    order = hdist.broadcast_ints(order, self.group, self.device)
            order = tuple(int(i) for i in order)
            if sorted(order) != list(range(n_act)):
                raise ValueError("order must be a permutation of range(%d)" % n_act)
        else:
            order = ()
        # ---- one pinned block: rows | noise | scalars of the Adam steps this update will take
        global_batch = int(global_batch) if global_batch is not None else B * self.world_size
        if global_batch < B:
            raise ValueError("global_batch %d is smaller than this rank's %d rows" % (global_batch, B))
        # Without noise from the caller, an update that fetches its own staging block (below) draws the noise on the
        # device, in that same launch, as the reference does (torch.randn_like on the device, algorithm.py:790, 1333):
        # the stream of hhrl_randn_fill for this learner's seed.  4096 draws cost the host 26 us per update otherwise.
        on_device = noise is None and self.group is None and self.use_graph is True
        key = (B, policy, order, gather, global_batch, on_device)
        ring = st.ring(key, W if gather else 0)
        slot = ring.next_slot()
        h = ring.h_np[slot]
        rows_on_device = gather or (isinstance(rows, torch.Tensor) and rows.is_cuda)
        if gather:
            ring.h_idx[slot][:] = _indices
        elif not rows_on_device:
            h[:st.n_rows].reshape(B, W)[:] = rows.numpy() if isinstance(rows, torch.Tensor) else rows
        lo = st.n_rows if rows_on_device else 0             # rows already on the device: noise, scalars and row numbers only
        n_up = st.total - lo
        in_body = self.group is None and self.use_graph is True and ring.zero_copy and n_up <= self.FETCH_MAX_FLOATS
        on_device = on_device and in_body
        for k, a in enumerate(fam.a_dims):
            if on_device:
                break
            dst = ring.h_noise[slot][k]
            if noise is None:
                self._np_rng.standard_normal(dtype=np.float32, out=dst)        # the draws of standard_normal((B, a)), in place
            else:
                src = noise[k].detach().cpu().numpy() if isinstance(noise[k], torch.Tensor) else np.asarray(noise[k])
                if tuple(src.shape) != (B, a):
                    raise ValueError("noise must be [%d, %d], got %s" % (B, a, tuple(src.shape)))
                dst[:] = src
        hyper = ring.h_hyper[slot]
        hyper[0][:] = self.critic.opt.next_step_scalars()
        if policy:
            for j, a in enumerate(self.actors):
                hyper[1 + j][:] = a.opt.next_step_scalars()
        if rows_on_device and not gather:
            st.rows.copy_(rows)
        # A small block is fetched by the update itself: a copy kernel that reads the pinned block in place is the first
        # launch of the chain (the first node of its graph; one graph per block).  A transfer enqueued between two
        # replays costs the device 14 us per update (WorkerTD3 at batch 4096: 95.8 -> 82.0 us; the stream hands over to
        # the copy engine and back, or, with a copy kernel on the stream, from one replay to a plain dispatch and on to
        # the next replay).  That holds up to blocks of megabytes (rows gathered on the host: HATD3Triple at batch 64, 256,
        # 1024 with 0.26, 1.0, 4.1 MB: 377 -> 361, 540 -> 524, 1213 -> 1200 us per update); beyond FETCH_MAX_FLOATS the
        # DMA transfer stays.
        stream = _lib.current_stream()
        scatter = None
        if gather and in_body:
            hdr = ring.h_new_hdr[slot]
            n_new = 0 if _new_rows is None else len(_new_rows)
            hdr[0] = n_new
            if n_new:
                hdr[1:1 + n_new] = _new_rows
                ring.h_new_rows[slot][:n_new] = self.memory[_new_rows]
                self._synced = self.pointer
            scatter = ring.h_ptr[slot] + 4 * ring.new_off
        elif _new_rows is not None:
            self._sync_replay()
        if in_body:
            hp, n0, n1 = ring.h_ptr[slot], st.noise_off[0], st.hyper_off[0]
            if on_device:                                   # everything but the noise, which the launch draws itself
                ring.h_rng[slot][0], ring.h_rng[slot][1] = self._noise_seed, self._noise_drawn
                self._noise_drawn += n1 - n0
                pieces = ((hp + 4 * n1, P(st.d_in, n1), st.total - n1),)
                if lo < n0:
                    pieces += ((hp + 4 * lo, P(st.d_in, lo), n0 - lo),)
                st.upload = (pieces, slot, (P(st.d_in, n0), n1 - n0, hp + 4 * ring.rng_off), scatter)
            else:
                st.upload = (((hp + 4 * lo, P(st.d_in, lo), n_up),), slot, None, scatter)
        else:
            st.upload = None
            st.d_in[lo:].copy_(ring.h_in[slot][lo:st.total], non_blocking=True)
            ring.release_after(slot, stream)
        st.gather = gather                                  # the update starts by gathering its rows on the device
        st.global_batch = global_batch
        self._run_body(key, st, B, policy, order)
        if in_body:
            ring.release_after(slot, stream)
        # host-side bookkeeping of what was just enqueued
        self.critic.opt.commit_step()
        self.critic_sched.step()
        if policy:
            for a in self.actors:
                a.opt.commit_step()
            for s in self.actor_scheds:
                s.step()
        self.last_critic_loss = self._loss

    def launch_forms(self) -> Dict[tuple, str]:
        """(batch, policy step?, actor order) -> "graph", "eager" or "tuning" for every launch chain seen
        so far; a benchmark warms up until none is "tuning" and every chain it expects has come up."""
        out = {}
        for key in self._seen:
            if key in self._slot_graphs:
                form = "graph" if self._slot_graphs[key] >= TD3Learner._Ring.SLOTS else "tuning"
            elif key in self._graphs or key in self._seg_graphs:
                form = "graph"
            elif self.use_graph is False:
                form = "eager"
            elif self.group is not None:
                form = "tuning"                 # captured after two eager runs
            elif self.use_graph is True:
                form = "tuning"
            else:
                form = "eager" if self._tune.get(key, {}).get("done") else "tuning"
            if form == "graph" and self.group is None and self.use_graph == "auto" and not self._tune[key]["done"]:
                form = "tuning"
            out[key] = form
        return out

    def _run_body(self, key, st, B: int, policy: bool, order: Sequence[int]) -> None:
        """Enqueues the update's kernels eagerly or as a captured HIP graph.

        use_graph=True (default): captured after two eager runs of a key, replayed from then on.  False:
        always eager.  "auto": the third and fourth runs of a key are timed eagerly, the graph is captured
        and its first two replays are timed the same way (a device synchronisation on either side), and
        the form with the shorter enqueue-to-finished time of ONE update is kept.  Back-to-back updates
        favour the graph more than that single-update latency shows (HATD3Triple: 0.68 ms replayed against
        0.76 ms eager in a loop, equal within noise when timed one at a time), hence the default."""
        if self.group is not None:
            return self._run_body_group(key, st, B, policy, order)
        # (a chain that fetches its own staging block has one graph per block: the block's address is part of the graph)
        gkey = key if st.upload is None else key + (st.upload[1],)
        graph = self._graphs.get(gkey)
        mode = self.use_graph
        tune = self._tune.get(key) if mode == "auto" else None
        if mode == "auto" and tune is None:
            tune = self._tune[key] = {"eager": [], "graph": [], "done": False}
        timed = tune is not None and not tune["done"] and self._seen.get(key, 0) >= 2
        if timed:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        if graph is not None:
            graph.replay()
            form = "graph"
        elif mode and self._seen.get(key, 0) >= (4 if mode == "auto" else 2) and not (tune is not None and tune["done"]):
            graph = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(graph):
                self._update_body(st, B, policy, order)
            self._graphs[gkey] = graph
            if gkey is not key:
                self._slot_graphs[key] = self._slot_graphs.get(key, 0) + 1
            if timed:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            graph.replay()
            form = "graph"
        else:
            self._seen[key] = self._seen.get(key, 0) + 1
            self._update_body(st, B, policy, order)
            form = "eager"
        if timed:
            torch.cuda.synchronize()
            tune[form].append(time.perf_counter() - t0)
            if len(tune["graph"]) >= 2:
                tune["done"] = True
                if min(tune["graph"]) > min(tune["eager"]):
                    del self._graphs[key]                     # eager is faster for this launch chain

    def _run_body_group(self, key, st, B: int, policy: bool, order: Sequence[int]) -> None:
        """Data-parallel form of _run_body.  An update has one gradient all-reduce per optimizer step
        (critic, then every actor of a policy step); the kernels between two all-reduces are captured as a
        HIP graph of their own after two eager runs of the launch chain, and an update is then
        graph, all-reduce, graph, ... (2 to 5 replays and 1 to 4 collectives instead of up to ~120 launches).
        The collectives stay outside the graphs: they belong to the process group's own stream handling."""
        from . import dist as hdist
        if self.graph_collectives:
            graph = self._graphs.get(key)
            if graph is None:
                if not self.use_graph or self._seen.get(key, 0) < 2:        # the eager runs also set the communicator up
                    self._seen[key] = self._seen.get(key, 0) + 1
                    self._update_body(st, B, policy, order)
                    return
                graph = torch.cuda.CUDAGraph()
                torch.cuda.synchronize()
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    self._update_body(st, B, policy, order)                  # _allreduce issues the collective into the capture
                self._graphs[key] = graph
            graph.replay()
            return
        segs = self._seg_graphs.get(key)
        if segs is None:
            if not self.use_graph or self._seen.get(key, 0) < 2:
                self._seen[key] = self._seen.get(key, 0) + 1
                self._update_body(st, B, policy, order)
                return
            segs = self._capture_segments(st, B, policy, order)
            self._seg_graphs[key] = segs
        dbg = _DP_DEBUG
        for n, (graph, flat) in enumerate(segs):
            graph.replay()
            if dbg:
                torch.cuda.synchronize()
                print("[dp rank %d] it %d key %s segment %d/%d replayed" % (self.rank, self.total_it, key[1:3], n, len(segs)), flush=True)
            if flat is not None:
                hdist.allreduce_sum_(flat, self.group)
                if dbg:
                    torch.cuda.synchronize()
                    print("[dp rank %d] it %d segment %d all-reduced %d floats" % (self.rank, self.total_it, n, flat.numel()), flush=True)

    def _capture_segments(self, st, B: int, policy: bool, order: Sequence[int]) -> list:
        """Runs _update_body under stream capture, cutting the capture at every all-reduce point:
        [(graph, flat gradient to all-reduce after it, or None for the last stretch)].  Nothing executes."""
        segs: list = []
        stream = torch.cuda.Stream(device=self.device)
        pool = torch.cuda.graph_pool_handle()
        cur = {}

        def begin():
            cur["graph"] = torch.cuda.CUDAGraph()
            # thread-local capture mode: the process group's helper threads may still be finishing HIP calls of
            # the previous collective (event queries, pinned-buffer bookkeeping), which the default global mode
            # would count as violations of this thread's capture
            cur["ctx"] = torch.cuda.graph(cur["graph"], pool=pool, stream=stream, capture_error_mode="thread_local")
            cur["ctx"].__enter__()

        def cut(flat):
            cur["ctx"].__exit__(None, None, None)
            segs.append((cur["graph"], flat))
            if flat is not None:
                begin()

        torch.cuda.synchronize()
        self._sync_hook = cut
        try:
            begin()
            try:
                self._update_body(st, B, policy, order)
            except BaseException:
                cur["ctx"].__exit__(None, None, None)      # never leave the stream capturing
                raise
            cut(None)
        finally:
            self._sync_hook = None
        return segs

    def _update_body(self, st: "TD3Learner._Static", B: int, policy: bool, order: Sequence[int]) -> None:
        """The kernel sequence of one update (no host-side state changes)."""
        fam, lib, sp = self.fam, self._lib, _lib.stream_ptr()
        W = fam.row_width
        rows, noise, pi, y = st.rows, st.noise, st.pi, st.y
        gather = bool(getattr(st, "gather", False))
        inv_n = 1.0 / float(getattr(st, "global_batch", B * self.world_size))
        R = lambda col: P(rows, col)
        goal = R(self.col_goal) if fam.goal_dim else None
        next_goal = R(self.col_next_goal) if fam.goal_dim else None
        obs = [(R(c), W) for c in self.col_obs]
        next_obs = [(R(c), W) for c in self.col_next_obs]

        # -- target value (no gradient): algorithm.py:1329-1337 / 785-796
        ct, cw_t = self.critic_target, self.critic_target.work(B)

        n_str = len(fam.streams)
        c, cw = self.critic, self.critic.work(B)
        # every LSTM that depends only on the minibatch goes into one grouped evaluation: the target actors'
        # and the target critic's on the next observations, the critic's on the current ones
        # (the target networks' evaluations are never back-propagated, algorithm.py:785: their steps are not saved)
        items = [a.lstm_item(B, next_obs[k][0], W, True) for k, a in enumerate(self.actor_targets) if a.lstm is not None]
        items += [ct.feature_item(k, B, next_obs[k][0], W, True) for k in range(n_str)]
        items += [c.feature_item(k, B, obs[k][0], W) for k in range(n_str)]
        # every piece of a network input that is a plain column range of the sampled rows: one launch -- with a
        # device-resident replay buffer the launch that gathers the rows themselves (the copies read the buffer)
        # ... and, where the update fetches its own staging block (rows | noise | Adam scalars | row numbers, st.upload),
        # the launch that copies the block out of pinned host memory: the first node of the update's graph
        cq = ops.CopyQueue(B, (P(rows), W, P(self._dev_mem), self.memory_capacity, st.idx.data_ptr()) if gather else None,
                           fetch=st.upload[0] if st.upload is not None else None)
        if st.upload is not None and st.upload[2] is not None:
            cq.add_normals(*st.upload[2])                # the smoothing noise of every actor, one run of the stream
        if st.upload is not None and st.upload[3] is not None:
            cq.add_scatter(st.upload[3], P(self._dev_mem), W, W)     # rows stored since the last update, into the mirror
        for k, a in enumerate(self.actor_targets):
            a.stage_inputs(cq, B, next_obs[k][0], W, next_goal, W)
        ct.stage_inputs(cq, B, next_obs, next_goal, W)
        c.stage_inputs(cq, B, obs, goal, W, R(self.col_act), W)
        # A single fused-MLP actor without an LSTM (the worker) does not depend on the critic step: on a policy step
        # its forward pass rides in the first launch and writes its action into a head input of its own
        # (state columns copied here), instead of taking a launch after the critic's optimizer step.
        ahead = (policy and len(self.actors) == 1 and self.actors[0].mlp3 is not None and self.actors[0].lstm is None
                 and c.mlp3 is not None and all(a.mlp3 is not None for a in self.actor_targets) and len(self.actor_targets) + 3 <= 8)
        if policy:
            for j, a in enumerate(self.actors):
                a.stage_inputs(cq, B, obs[j][0], W, goal, W)
            if ahead:
                c.stage_inputs(cq, B, obs, goal, W, policy_input=True)
        # zero_grad of the critic (algorithm.py:805) and, on a policy step, of every actor (:822; nothing touches an
        # actor's gradient before its own backward pass) ride in the same launch
        cq.add_zero(c.opt.g.base_ptr, c.opt.g.data.numel())
        cq.add_zero(P(self._loss), 2)               # the two critic losses are accumulated by the launch that forms them
        if policy:
            for a in self.actors:
                cq.add_zero(a.opt.g.base_ptr, a.opt.g.data.numel())
        cq.flush()
        self._lstm_forward(items, B)

        # target policies: layer k of every actor in one launch, then the smoothing noise
        for k, a in enumerate(self.actor_targets):
            a.bind_input(B, next_obs[k][0], W)
        # with fused three-layer networks throughout, the critic's own heads on the replayed (s, a) -- they depend
        # on the minibatch only -- ride in the target policies' launch instead of taking one of their own below
        n_a = len(self.actors)
        heads_early = c.mlp3 is not None and all(a.mlp3 is not None for a in self.actor_targets) and n_a + 2 <= 8
        smoothed = False
        if heads_early:
            d = ops.mlp3_descs(n_a + 2 + (1 if ahead else 0))
            for k, a in enumerate(self.actor_targets):
                a.fused_desc(d[k], B, P(pi[k]), fam.a_dims[k])
                # the smoothing noise is applied in the launch's epilogue, straight into the target critic's input
                d[k].smooth_eps, d[k].y2, d[k].ldy2 = P(noise[k]), P(cw_t.xu, ct.act_cols[k]), ct.D
            c.head_descs(d, n_a, B)
            if ahead:
                self.actors[0].bind_input(B, obs[0][0], W)
                self.actors[0].fused_desc(d[n_a + 2], B, P(cw.xu_pi, c.act_cols[0]), c.D)
            ops.mlp3_forward_group(d, B)
            smoothed = True
        elif c.mlp3 is None and all(a.mlp3 is None for a in self.actor_targets):
            # layer by layer: layer k of every target actor and of both heads of the critic in one launch
            heads_early = True
            for level in (1, 2, 3):
                items = [a.level_item(level, B, P(pi[k]), fam.a_dims[k]) for k, a in enumerate(self.actor_targets)]
                if level == 3:              # the smoothing noise in the last layer's epilogue, straight into the target critic's input
                    items = [it + (P(noise[k]), P(cw_t.xu, ct.act_cols[k]), ct.D) for k, it in enumerate(items)]
                    smoothed = True
                ops.linear_forward_group(items + c.level_items(level, B), B)
        else:
            self._actors_forward(self.actor_targets, B, [(P(pi[k]), fam.a_dims[k]) for k in range(n_a)])
        for k in range(len(self.actors) if not smoothed else 0):
            check(lib.hhrl_td3_smooth_action(P(pi[k]), fam.a_dims[k], P(noise[k]), P(cw_t.xu, ct.act_cols[k]), ct.D, B,
                                             fam.a_dims[k], sp))
        ct.heads(B)

        # -- critic step: :1341-1350 / 799-808; target value, both losses and their gradients in one launch
        if not heads_early:
            c.heads(B)
        # (fused three-layer heads form the target, the losses and dL/dQ inside their backward launch)
        c_target = None
        if c.fused_target():
            c_target = (P(cw_t.q1), P(cw_t.q2), R(self.col_reward), W, fam.gamma, inv_n, P(self._loss), P(y))
        else:
            check(lib.hhrl_td3_target_mse_zeroed(P(cw_t.q1), P(cw_t.q2), R(self.col_reward), W, fam.gamma, P(cw.q1), P(cw.q2), inv_n,
                                                 P(y), P(self._loss), P(cw.d_q1), P(cw.d_q2), B, sp))
        # on a policy step the soft updates (:1361-1364 / 827-834) ride on the optimizer launches: nothing
        # between a network's step and the end of the update reads its target.  On one device (no gradient
        # exchange between backward pass and step) a fused network's step rides on its backward call in turn.
        local = self.world_size == 1
        c_ride = (P(st.hyper[0]), self.critic_target.params if policy else None, fam.tau)
        stepped = c.backward(B, both_heads=True, param_grads=True, lstms=False, ride=c_ride if local else None, target=c_target)
        self._lstm_backward([c.lstm_bwd_item(k, B) for k in range(n_str)], B)
        self._allreduce(c.opt.g)
        if not stepped:
            c.opt.step_dev(*c_ride)

        # -- delayed policy step: :1353-1366 / 811-838
        if policy:
            # LSTM features under the updated critic weights, and every actor's LSTM, in one grouped evaluation
            # (the policy step differentiates Q1 with respect to the actions only: the critic's LSTMs save nothing)
            items = [c.feature_item(k, B, obs[k][0], W, True) for k in range(n_str)]
            items += [a.lstm_item(B, obs[j][0], W) for j, a in enumerate(self.actors) if a.lstm is not None]
            self._lstm_forward(items, B)
            # joint action written straight into the critic input: layer k of every actor in one launch
            xu_act = cw.xu_pi if ahead else cw.xu
            if not ahead:
                for j, a in enumerate(self.actors):
                    a.bind_input(B, obs[j][0], W)
                self._actors_forward(self.actors, B, [(P(cw.xu, c.act_cols[j]), c.D) for j in range(len(self.actors))])
            for n, i in enumerate(order):
                a = self.actors[i]
                if not c.q1_input_gradient(B, -inv_n, policy_input=ahead):                                     # d(-mean Q1)/d(s, a)
                    c.heads(B, q1_only=True, policy_input=ahead, hidden_only=True)
                    c.backward(B, both_heads=False, param_grads=False, dq_const=-inv_n, policy_input=ahead,
                               dx_cols=(c.act_cols[i], fam.a_dims[i]))
                a_ride = (P(st.hyper[1 + i]), self.actor_targets[i].params, fam.tau)
                stepped = a.backward(B, P(xu_act, c.act_cols[i]), c.D, P(cw.d_xu, c.act_cols[i]), c.D, lstm=False,
                                     ride=a_ride if local else None)
                if a.lstm is not None:
                    self._lstm_backward([a.lstm_bwd_item(B)], B)
                self._allreduce(a.opt.g)
                if not stepped:
                    a.opt.step_dev(*a_ride)
                if n + 1 < len(order):                # later actors see this actor's new policy
                    a.forward(B, obs[i][0], W, goal, W, P(cw.xu, c.act_cols[i]), c.D, inputs_done=True, no_backward=True)

    @staticmethod
    def _actors_forward(nets, B: int, outs) -> None:
        """The MLPs of several actors on bound inputs: one fused launch when every network is covered
        by the three-layer kernels, else layer k of every actor per launch."""
        if all(a.mlp3 is not None for a in nets):
            d = ops.mlp3_descs(len(nets))
            for k, a in enumerate(nets):
                a.fused_desc(d[k], B, outs[k][0], outs[k][1])
            ops.mlp3_forward_group(d, B)
            return
        for level in (1, 2, 3):
            ops.linear_forward_group([a.level_item(level, B, outs[k][0], outs[k][1]) for k, a in enumerate(nets)], B)

    @staticmethod
    def _by_shape(items):
        groups: Dict[Tuple[int, int], list] = {}
        for it in items:
            if it is not None:
                groups.setdefault((it[0].T, it[0].H), []).append(it)
        return groups.values()

    def _lstm_forward(self, items, B: int) -> None:
        for grp in self._by_shape(items):
            ops.lstm_forward_group(grp, B)

    def _lstm_backward(self, items, B: int) -> None:
        for grp in self._by_shape(items):
            ops.lstm_backward_group(grp, B)

    def _allreduce(self, g: FlatBuffer) -> None:
        if self.group is not None:
            if self._sync_hook is not None:         # under capture: the collective runs between the graphs
                self._sync_hook(g.data)
                return
            from . import dist as hdist
            hdist.allreduce_sum_(g.data, self.group)

    def _learn(self, indices=None, noise=None, order=None, global_batch=None) -> None:
        if indices is None:
            # np.random.choice(n, size=B) (algorithm.py:286, 482, 771, 1319): with replacement and no weights that IS
            # randint(0, n, size=B) -- same values, same generator state afterwards -- without 2 us of argument checks
            indices = np.random.randint(0, min(self._n_stored(), self.memory_capacity), size=self.batch_size)
        if self.device_replay and hasattr(self, "_sync_replay"):
            # Rows stored since the last update (one per control step, typically) ride in the update's staging block
            # instead of a transfer of their own in front of it -- unless this minibatch samples one of them (the gather
            # and the scatter share a launch), or there are many (bulk fills, mark_replay_dirty()).
            indices = np.asarray(indices)
            cap = self.memory_capacity
            new = min(self.pointer - self._synced, cap)
            ride = None
            if 0 < new <= TD3Learner._Ring.NEW_ROWS_MAX and self.group is None and self.use_graph is True:
                first = (self.pointer - new) % cap
                if not ((indices - first) % cap < new).any():
                    ride = (first + np.arange(new)) % cap
            if ride is None:
                self._sync_replay()
            self.learn_rows(None, noise, order, _indices=indices, global_batch=global_batch, _new_rows=ride)
        else:
            self.learn_rows(self._sample_rows(np.asarray(indices)), noise, order, global_batch=global_batch)

    # ---- inference --------------------------------------------------------
    def _act(self, k: int, obs, goal=None) -> np.ndarray:
        """Batch-1 policy evaluation (choose_* methods, algorithm.py:279-281 ...)."""
        s = self.fam.streams[k]
        o = np.asarray(obs, dtype=np.float32).reshape(1, -1)
        if o.shape[1] != s.width:
            raise ValueError("expected %d observation values, got %d" % (s.width, o.shape[1]))
        return self._policy_batch(k, o, goal)[0]

    def _policy_batch(self, k: int, obs, goal=None) -> np.ndarray:
        """Actor k on n observations (and goals): one pinned block up, one launch chain, one pinned block down.
        An actor with an LSTM is a chain of T + 5 launches: after two eager calls for a given n it is
        captured into a HIP graph and replayed (unless use_graph=False)."""
        fam, s = self.fam, self.fam.streams[k]
        o = np.asarray(obs, dtype=np.float32).reshape(-1, s.width)
        n, gd, a = int(o.shape[0]), fam.goal_dim, fam.a_dims[k]
        key = (k, n)
        io = self._policy_io.get(key)
        if io is None:
            if len(self._policy_io) > 32:
                self._policy_io.clear()
            wid = s.width + gd
            h_in = torch.zeros(n, wid, dtype=torch.float32).pin_memory()
            h_out = torch.zeros(n, a, dtype=torch.float32).pin_memory()
            # small blocks are read and written by the kernels in place (no staging copies: one launch or replay per call)
            zc = _lib.zero_copy_ok(h_in) and _lib.zero_copy_ok(h_out)
            io = self._policy_io[key] = {"h_in": h_in, "h_out": h_out, "zc": zc,
                                         "d_in": h_in if zc else ops.dev_empty(n, wid, device=self.device),
                                         "d_out": h_out if zc else ops.dev_empty(n, a, device=self.device),
                                         "calls": 0, "graph": None}
        h_in, d_in, d_out, h_out, zc = io["h_in"], io["d_in"], io["d_out"], io["h_out"], io["zc"]
        wid = s.width + gd
        hv = h_in.numpy()
        hv[:, :s.width] = o
        if gd:
            hv[:, s.width:] = np.asarray(goal, dtype=np.float32).reshape(n, gd)
        if not zc:
            d_in.copy_(h_in, non_blocking=True)
        run = lambda: self.actors[k].forward(n, P(d_in), wid, P(d_in, s.width) if gd else None, wid, P(d_out), a, no_backward=True)
        if io["graph"] is not None:
            io["graph"].replay()
        elif self.use_graph and self.actors[k].lstm is not None and io["calls"] >= 2:
            graph = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(graph):
                run()
            io["graph"] = graph
            graph.replay()
        else:
            io["calls"] += 1
            run()
        if not zc:
            h_out.copy_(d_out, non_blocking=True)
        _lib.current_stream().synchronize()        # the pinned blocks are free again afterwards
        return h_out.numpy().copy()

    # ---- checkpoints ------------------------------------------------------
    def state_dicts(self) -> Dict[str, Dict[str, torch.Tensor]]:
        d = {n: a.params.state_dict() for n, a in zip(self.actor_names, self.actors)}
        d["critic"] = self.critic.params.state_dict()
        return d

    def target_state_dicts(self) -> Dict[str, Dict[str, torch.Tensor]]:
        d = {n: a.params.state_dict() for n, a in zip(self.actor_names, self.actor_targets)}
        d["critic"] = self.critic_target.params.state_dict()
        return d

    def load_state_dicts(self, sds: Dict[str, dict]) -> None:
        """Loads networks and re-creates the targets as copies, like `load`
        does with deepcopy (algorithm.py:337-344)."""
        for n, a, t in zip(self.actor_names, self.actors, self.actor_targets):
            if n in sds:
                a.params.load(sds[n])
                t.params.copy_from(a.params)
        if "critic" in sds:
            self.critic.params.load(sds["critic"])
            self.critic_target.params.copy_from(self.critic.params)

    def _nets(self):
        return list(zip(self.actor_names, self.actors)) + [("critic", self.critic)]

    def save(self, filename: str) -> None:
        """`<prefix>_actor[k]`, `_actor[k]_optimizer`, `_critic`, `_critic_optimizer`
        (algorithm.py:330-335, 544-551, 840-849, 1170-1175, 1368-1373)."""
        for n, net in self._nets():
            torch.save(net.params.state_dict(), filename + "_" + n)
            torch.save(net.opt.state_dict(), filename + "_" + n + "_optimizer")

    def load(self, filename: str) -> None:
        sds = {}
        for n, net in self._nets():
            sds[n] = torch.load(filename + "_" + n, weights_only=True)
            net.opt.load_state_dict(torch.load(filename + "_" + n + "_optimizer", weights_only=True))
        self.load_state_dicts(sds)


class _FlatReplayMixin:
    """`np.zeros((capacity, row))` float64 ring buffer with np.hstack rows
    (algorithm.py:252, 273-277)."""

    def _init_memory(self):
        self.memory = np.zeros((self.memory_capacity, self.fam.row_width))
        self.pointer = 0
        if self.device_replay:
            self._dev_mem = ops.dev_zeros(self.memory_capacity, self.fam.row_width, device=self.device)
            self._synced = 0                         # value of `pointer` up to which the mirror is current
            n_stage = min(self.memory_capacity, 4096)
            self._stage = torch.zeros(n_stage, self.fam.row_width, dtype=torch.float32).pin_memory()
            self._stage_np = self._stage.numpy()
            self._stage_ev = None

    def mark_replay_dirty(self) -> None:
        """Call after writing `memory` rows directly that were already mirrored (store_transition and
        writes followed by a larger `pointer` are picked up without it): the whole buffer is uploaded again."""
        if self.device_replay:
            self._synced = self.pointer - self.memory_capacity

    def _sync_replay(self) -> None:
        """Uploads the rows stored since the last update into the device mirror (fp32), oldest first."""
        cap = self.memory_capacity
        new = min(self.pointer - self._synced, cap)
        if new <= 0:
            return
        lo = (self.pointer - new) % cap
        spans = [(lo, min(lo + new, cap))]
        if lo + new > cap:
            spans.append((0, lo + new - cap))
        n_stage = self._stage.shape[0]
        for a, b in spans:
            for c in range(a, b, n_stage):
                n = min(n_stage, b - c)
                if self._stage_ev is not None:
                    self._stage_ev.synchronize()            # the previous transfer has read the staging block
                self._stage_np[:n] = self.memory[c:c + n]
                self._dev_mem[c:c + n].copy_(self._stage[:n], non_blocking=True)
                self._stage_ev = torch.cuda.Event()
                self._stage_ev.record()
        self._synced = self.pointer

    def _store(self, *parts) -> None:
        transition = np.hstack(parts)
        self.memory[self.pointer % self.memory_capacity, :] = transition
        self.pointer += 1

    def _sample_rows(self, indices):
        return self.memory[indices, :]

    def _n_stored(self):
        return self.pointer


# ------------------------------------------------------------------ reference surface
class TD3Single(_FlatReplayMixin, TD3Learner):
    """algorithm.py:235-344."""

    def __init__(self, cfg: dict, task: str, **kw):
        super().__init__(family_from_cfg("single", cfg, task), **kw)
        s = self.fam.streams[0]
        self.o_dim, self.a_dim, self.s_dim, self.t = s.o, self.fam.a_dims[0], s.s, s.T
        self.obs_dim = s.width
        self.a1_dim = self.a2_dim = self.a3_dim = self.a_dim
        self.choose_time_action = self.choose_phase_action = self.choose_goal = self.choose_action
        self._init_memory()

    def store_transition(self, o, a, r, o_):
        self._store(o, a, r, o_)

    def choose_action(self, o):
        return self._act(0, o).flatten()

    def learn(self, indices=None, noise=None, order=None, global_batch=None):
        self._learn(indices, noise, order, global_batch)


class HATD3Double(_FlatReplayMixin, TD3Learner):
    """algorithm.py:409-563 (time|phase + vehicle)."""

    _kind = "double"

    def __init__(self, cfg: dict, light_opt: Optional[str] = None, **kw):
        super().__init__(family_from_cfg(self._kind, cfg, light_opt), **kw)
        self.use_opc = cfg["use_opc"]
        s1, s2 = self.fam.streams
        self.o1_dim, self.s1_dim, self.a1_dim, self.t1 = s1.o, s1.s, self.fam.a_dims[0], s1.T
        self.o2_dim, self.s2_dim, self.a2_dim, self.t2 = s2.o, s2.s, self.fam.a_dims[1], s2.T
        self.obs1_dim, self.obs2_dim = s1.width, s2.width
        self.obs_dim, self.act_dim = self.fam.obs_dim, self.fam.act_dim
        self.choose_time_action = self.choose_phase_action = self.choose_light_action
        self._init_memory()

    def store_transition(self, o1, o2, a1, a2, r, o1_, o2_):
        self._store(o1, o2, a1, a2, r, o1_, o2_)

    def choose_light_action(self, o1):
        return self._act(0, o1).flatten()

    def choose_goal(self, o2):
        return self._act(1, o2).flatten()

    def choose_goal_batch(self, o2_rows):
        """choose_goal for several observation windows at once (the eight lanes of an intersection at a
        decision point, agent.py:405-419) -> [n, a2_dim]: one launch chain and one host round trip."""
        return self._policy_batch(1, o2_rows)

    def learn(self, worker_policy=None, indices=None, noise=None, order=None, global_batch=None):
        self._learn(indices, noise, order, global_batch)


class HATD3(HATD3Double):
    """algorithm.py:567-606 (time + phase)."""

    _kind = "tp"

    def __init__(self, cfg: dict, **kw):
        super().__init__(cfg, None, **kw)
        self.choose_time_action = self.choose_light_action
        self.choose_phase_action = self.choose_goal


class HATD3Triple(_FlatReplayMixin, TD3Learner):
    """algorithm.py:679-864 (time + phase + vehicle)."""

    def __init__(self, cfg: dict, **kw):
        super().__init__(family_from_cfg("triple", cfg), **kw)
        self.use_opc = cfg["use_opc"]
        for i, s in enumerate(self.fam.streams, 1):
            setattr(self, "o%d_dim" % i, s.o)
            setattr(self, "s%d_dim" % i, s.s)
            setattr(self, "a%d_dim" % i, self.fam.a_dims[i - 1])
            setattr(self, "t%d" % i, s.T)
            setattr(self, "obs%d_dim" % i, s.width)
        self.obs_dim, self.act_dim = self.fam.obs_dim, self.fam.act_dim
        self._init_memory()

    def store_transition(self, o1, o2, o3, a1, a2, a3, r, o1_, o2_, o3_):
        self._store(o1, o2, o3, a1, a2, a3, r, o1_, o2_, o3_)

    def choose_time_action(self, o1):
        return self._act(0, o1).flatten()

    def choose_phase_action(self, o2):
        return self._act(1, o2).flatten()

    def choose_goal(self, o3):
        return self._act(2, o3).flatten()

    def choose_goal_batch(self, o3_rows):
        """choose_goal for several observation windows at once (the eight lanes of an intersection at a
        decision point, agent.py:405-419) -> [n, a3_dim]: one launch chain and one host round trip."""
        return self._policy_batch(2, o3_rows)

    def learn(self, worker_policy=None, indices=None, noise=None, order=None, global_batch=None):
        self._learn(indices, noise, order, global_batch)


class ManagerTD3(TD3Learner):
    """algorithm.py:915-1184 (goal-only manager), including the HIRO off-policy correction of the stored
    goals (`use_opc`, :1007-1107, 1128-1132; switched off in the reference's configuration, configs.py:73)."""

    def __init__(self, cfg: dict, **kw):
        super().__init__(family_from_cfg("manager", cfg), **kw)
        self.use_opc = bool(cfg["use_opc"])
        s = self.fam.streams[0]
        self.o_dim, self.a_dim, self.s_dim, self.t = s.o, self.fam.a_dims[0], s.s, s.T
        self.obs_dim = s.width
        self.choose_goal = self.choose_action
        self.memory = DictReplay(self.memory_capacity, self.obs_dim, self.a_dim, 25, self.use_opc)
        self.store_transition = self.memory.store_transition
        self._worker_policy = None

    @property
    def pointer(self):
        return self.memory.pointer

    def _n_stored(self):
        return self.memory.pointer

    def _sample_rows(self, indices):
        m = self.memory
        action = m["action"][indices]
        if self.use_opc:
            if self._worker_policy is None:
                raise ValueError("use_opc=True: learn() needs the worker policy (algorithm.py:1109, 1130)")
            action = self.off_policy_corrections(self._worker_policy, action, m["worker_obs"][indices], m["worker_act"][indices])
        return np.hstack((m["obs"][indices], action, m["reward"][indices], m["next_obs"][indices]))

    @staticmethod
    def off_policy_corrections(worker_policy, sgoals, wstates, wactions, candidate_goals: int = 8) -> np.ndarray:
        """Re-labels the stored manager goals (algorithm.py:1007-1107): per lane, among the stored goal,
        the observed change of the goal feature and `candidate_goals` samples around it, the candidate
        under which the current worker policy best reproduces the stored worker actions.

        sgoals [B, L]; wstates [B, S, L, wo]; wactions [B, S, L, wa] -> corrected goals [B, L].
        The random candidates are drawn lane by lane from numpy's global generator, in the reference's
        order; the policy is then evaluated for every (lane, candidate, row, step) in ONE batched call
        instead of L * (candidate_goals + 2) calls."""
        st = np.transpose(np.asarray(wstates), (2, 0, 1, 3))             # [L, B, S, wo]
        act = np.transpose(np.asarray(wactions), (2, 0, 1, 3))           # [L, B, S, wa]
        goals = np.transpose(np.asarray(sgoals), (1, 0))                 # [L, B]
        L, B, S, wo = st.shape
        wa = act.shape[-1]
        gdim = 1                                                         # one goal value per lane (:1036)
        C = candidate_goals + 2
        cands = np.empty((L, B, C, gdim))
        for lane in range(L):
            diff = (st[lane][:, -1] - st[lane][:, 0])[:, None, -gdim:]                    # [B, 1, gdim]
            orig = goals[lane][:, None, None]
            rnd = np.random.normal(loc=diff, scale=.5, size=(B, candidate_goals, gdim)).clip(-1, 1)
            cands[lane] = np.concatenate([orig, diff, rnd], axis=1)
        # goal seen by the worker at step s under candidate c: (c + state_0) - state_s on the goal feature (:1073)
        start = st[:, :, 0, -gdim:]                                       # [L, B, gdim]
        cand_t = np.transpose(cands, (0, 2, 1, 3))                        # [L, C, B, gdim]
        goal_in = (cand_t + start[:, None])[:, :, :, None, :] - st[:, None, :, :, -gdim:]     # [L, C, B, S, gdim]
        obs_in = np.broadcast_to(st[:, None], (L, C, B, S, wo))
        flat_obs = np.ascontiguousarray(obs_in).reshape(-1, wo)
        pa = np.asarray(worker_policy(flat_obs, goal_in.reshape(-1, gdim)), dtype=np.float64).reshape(L, C, B, S, wa)
        # Steps without a controlled vehicle take action -1 instead of the policy's (:1084-1086).  The
        # reference tests `np.any` of the masked row NUMBERS of a lane, so a lane whose only masked
        # step is row 0 keeps the policy's action there; kept as is.
        mask = st[..., 0] == -1                                           # [L, B, S]
        mask &= mask.reshape(L, -1)[:, 1:].any(axis=1)[:, None, None]
        pa[np.broadcast_to(mask[:, None], (L, C, B, S))] = -1
        diff = pa - act[:, None]
        diff = np.where(diff != -np.inf, diff, 0)
        logprob = -0.5 * np.sum(np.linalg.norm(diff, axis=-1) ** 2, axis=-1)                 # [L, C, B]
        best = np.argmax(logprob, axis=1)                                 # [L, B]
        res = np.take_along_axis(cands, best[:, :, None, None], axis=2)[:, :, 0, 0]          # [L, B]
        return np.transpose(res, (1, 0))

    def choose_action(self, o_l):
        return self._act(0, o_l).flatten()

    def learn(self, worker_policy=None, indices=None, noise=None, order=None, global_batch=None):
        self._worker_policy = worker_policy
        try:
            self._learn(indices, noise, order, global_batch)
        finally:
            self._worker_policy = None


class WorkerTD3(_FlatReplayMixin, TD3Learner):
    """algorithm.py:1261-1382 (goal-conditioned CAV agent)."""

    def __init__(self, cfg: dict, **kw):
        super().__init__(family_from_cfg("worker", cfg), **kw)
        s = self.fam.streams[0]
        self.o_dim, self.a_dim, self.s_dim, self.t = s.o, self.fam.a_dims[0], s.s, s.T
        self.g_dim = self.fam.goal_dim
        self.obs_dim = s.width
        self._init_memory()

    def store_transition(self, o, a, g, g_, r, o_):
        self._store(o, a, g, g_, r, o_)

    def choose_action(self, o, g):
        return self._act(0, o, g).flatten()

    def policy(self, o, g):
        """Batched policy for the manager's off-policy correction (algorithm.py:1310-1314)."""
        return self._policy_batch(0, o, g)

    def learn(self, indices=None, noise=None, order=None, global_batch=None):
        self._learn(indices, noise, order, global_batch)
