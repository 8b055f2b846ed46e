"""Thin, allocation-free wrappers that enqueue the HIP kernels on torch's
current stream.  Operands are raw device addresses (ints) plus leading
dimensions, so column slices of wider buffers are addressed directly."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH, check
from .flat import FlatBuffer


def P(t: torch.Tensor, col: int = 0) -> int:
    """Device address of element `col` of a contiguous fp32 tensor."""
    return t.data_ptr() + 4 * col


def dev_empty(*shape, device) -> torch.Tensor:
    return torch.empty(*shape, dtype=torch.float32, device=device)


def dev_zeros(*shape, device) -> torch.Tensor:
    return torch.zeros(*shape, dtype=torch.float32, device=device)


class Lin:
    """One dense layer bound to its slices of the flat parameter / gradient
    buffers (nn.Linear layout: weight [out, in], bias [out])."""

    def __init__(self, params: FlatBuffer, grads: Optional[FlatBuffer], name: str, act: int = ACT_NONE):
        self.name = name
        self.N, self.K = params.layout.shapes[name + ".weight"]
        self.act = act
        self.w = params.ptr(name + ".weight")
        self.b = params.ptr(name + ".bias")
        self.gw = grads.ptr(name + ".weight") if grads is not None else None
        self.gb = grads.ptr(name + ".bias") if grads is not None else None
        self._lib = _lib.load()

    def fwd(self, x: int, ldx: int, y: int, ldy: int, M: int, accumulate: int = 0) -> None:
        check(self._lib.hhrl_linear_forward(x, ldx, self.w, self.b, y, ldy, M, self.K, self.N, self.act,
                                            accumulate, _lib.stream_ptr()))

    def bwd(self, x: Optional[int], ldx: int, y: Optional[int], ldy: int, dy: int, lddy: int,
            dx: Optional[int], lddx: int, M: int, dx_accumulate: int = 0, param_grads: bool = True) -> None:
        gw = self.gw if param_grads else None
        gb = self.gb if param_grads else None
        check(self._lib.hhrl_linear_backward(x, ldx, self.w, y, ldy, dy, lddy, dx, lddx, dx_accumulate,
                                             gw, gb, M, self.K, self.N, self.act, _lib.stream_ptr()))


def linear_backward_row(lin: "Lin", y: int, ldy: int, dy_row: int, alpha: float, dx: int, lddx: int, M: int,
                        dx_accumulate: int = 0) -> None:
    """dx = alpha * ((dy_row for every sample) * act'(y)) W of layer `lin`: its input gradient when dL/dy is one row."""
    check(_lib.load().hhrl_linear_backward_row(lin.w, y, ldy, dy_row, alpha, dx, lddx, dx_accumulate, M, lin.K, lin.N, lin.act,
                                               _lib.stream_ptr()))


def linear_backward_cols(lin: "Lin", y: int, ldy: int, dy: int, lddy: int, dx: int, lddx: int, c0: int, nc: int, M: int,
                         dx_accumulate: int = 0) -> None:
    """Columns [c0, c0 + nc) of layer `lin`'s input gradient (the others are left alone)."""
    check(_lib.load().hhrl_linear_backward_cols(lin.w, lin.K, y, ldy, dy, lddy, dx, lddx, dx_accumulate, c0, nc, M, lin.N, lin.act,
                                                _lib.stream_ptr()))


def linear_forward_group(items, M: int) -> None:
    """items: [(Lin, x, ldx, y, ldy[, smooth_eps, y2, ldy2])]: independent layers on the same M rows, one launch."""
    n = len(items)
    arr = (_lib.LinearFwdDesc * n)()
    for d, it in zip(arr, items):
        l, x, ldx, y, ldy = it[:5]
        d.x, d.ldx, d.w, d.b, d.y, d.ldy = x, ldx, l.w, l.b, y, ldy
        d.K, d.N, d.act, d.accumulate = l.K, l.N, l.act, 0
        if len(it) > 5:                 # (..., smooth_eps, y2, ldy2): target-policy smoothing in the launch's epilogue
            d.smooth_eps, d.y2, d.ldy2 = it[5:8]
    check(_lib.load().hhrl_linear_forward_group(arr, n, M, _lib.stream_ptr()))


def linear_backward_group(items, M: int, param_grads: bool = True) -> None:
    """items: [(Lin, x, ldx, y, ldy, dy, lddy, dx, lddx, dx_accumulate)] with the meaning of Lin.bwd;
    one launch per stage (activation/bias gradient, input gradients, weight gradients)."""
    n = len(items)
    arr = (_lib.LinearBwdDesc * n)()
    for d, (l, x, ldx, y, ldy, dy, lddy, dx, lddx, acc) in zip(arr, items):
        d.x, d.ldx, d.w, d.y, d.ldy, d.dy, d.lddy = x, ldx, l.w, y, ldy, dy, lddy
        d.dx, d.lddx, d.dx_accumulate = dx, lddx, acc
        d.dw = l.gw if param_grads else None
        d.db = l.gb if param_grads else None
        d.K, d.N, d.act = l.K, l.N, l.act
    check(_lib.load().hhrl_linear_backward_group(arr, n, M, _lib.stream_ptr()))


class Mlp3:
    """Three consecutive dense layers (ReLU, ReLU, act3) bound to the fused kernels."""

    def __init__(self, l1: Lin, l2: Lin, l3: Lin):
        self.l = (l1, l2, l3)
        self.K0, self.N3, self.act3 = l1.K, l3.N, l3.act
        self._slabs = {}          # rows M -> workspace for the slab hand-over of the parameter gradients (large M only)

    def slab(self, M: int, device) -> Optional[torch.Tensor]:
        """Workspace that lets the backward call sum its workgroups' parameter gradients in a fixed order
        instead of adding them with atomics."""
        if M not in self._slabs:
            n = _lib.load().hhrl_mlp3_slab_floats(self.K0, self.N3, M)
            self._slabs[M] = dev_empty(n, device=device) if n > 0 else None
        return self._slabs[M]

    @staticmethod
    def supported(l1: Lin, l2: Lin, l3: Lin) -> bool:
        if l1.act != ACT_RELU or l2.act != ACT_RELU or l2.K != l1.N or l3.K != l2.N:
            return False
        return bool(_lib.load().hhrl_mlp3_supported(l1.K, l1.N, l2.N, l3.N))

    def desc(self, d, x: int, ldx: int, y: int, ldy: int, h1: Optional[int], h2: Optional[int]) -> None:
        l1, l2, l3 = self.l
        d.x, d.ldx, d.K0 = x, ldx, self.K0
        d.w1, d.b1, d.w2, d.b2, d.w3, d.b3 = l1.w, l1.b, l2.w, l2.b, l3.w, l3.b
        d.N3, d.act3 = self.N3, self.act3
        d.h1, d.h2, d.y, d.ldy = h1, h2, y, ldy

    def grads(self, d, dy: Optional[int], lddy: int, dy_const: float, param_grads: bool, dx: Optional[int], lddx: int,
              slab: Optional[torch.Tensor] = None) -> None:
        l1, l2, l3 = self.l
        d.dy, d.lddy, d.dy_const = dy, lddy, dy_const
        if param_grads:
            d.gw1, d.gb1, d.gw2, d.gb2, d.gw3, d.gb3 = l1.gw, l1.gb, l2.gw, l2.gb, l3.gw, l3.gb
            if slab is not None:
                d.gslab, d.gslab_floats = slab.data_ptr(), slab.numel()
        d.dx, d.lddx = dx, lddx


def mlp3_descs(n: int):
    return (_lib.Mlp3Desc * n)()


def mlp3_forward_group(descs, M: int) -> None:
    check(_lib.load().hhrl_mlp3_forward_group(descs, len(descs), M, _lib.stream_ptr()))


def mlp3_backward_group(descs, M: int, ride=None) -> None:
    """ride = (FlatAdam, hyper_ptr, target FlatBuffer or None, tau): the optimizer's step (and the target's soft
    update) follows the gradients in the same call; the group's networks must be all of that optimizer's parameters."""
    if ride is None:
        check(_lib.load().hhrl_mlp3_backward_group(descs, len(descs), M, _lib.stream_ptr()))
        return
    opt, hyper, target, tau = ride
    r = _lib.AdamRide(opt.g.base_ptr, opt.p.base_ptr, opt.m.base_ptr, opt.v.base_ptr, opt.p.layout.size, hyper,
                      target.base_ptr if target is not None else None, float(tau))
    check(_lib.load().hhrl_mlp3_backward_group_adam(descs, len(descs), M, C.byref(r), _lib.stream_ptr()))


LSTM_KEYS = ["lstm_layer.weight_ih_l0", "lstm_layer.weight_hh_l0", "lstm_layer.bias_ih_l0", "lstm_layer.bias_hh_l0",
             "lstm_layer.weight_ih_l0_reverse", "lstm_layer.weight_hh_l0_reverse",
             "lstm_layer.bias_ih_l0_reverse", "lstm_layer.bias_hh_l0_reverse", "fc.weight", "fc.bias"]


class BiLstm:
    """The reference's LSTM module (algorithm.py:141-156) bound to its slices
    of the flat buffers; `prefix` is e.g. 'lstm.' or 'lstm2.'."""

    def __init__(self, params: FlatBuffer, grads: Optional[FlatBuffer], prefix: str, obs_dim: int, T: int, hid: int):
        self.o, self.T, self.H = obs_dim, T, hid
        self._lib = _lib.load()
        self.params = (C.c_void_p * 10)(*[params.ptr(prefix + k) for k in LSTM_KEYS])
        self.grads = (C.c_void_p * 10)(*[grads.ptr(prefix + k) for k in LSTM_KEYS]) if grads is not None else None

    def workspace(self, B: int, device) -> torch.Tensor:
        n = self._lib.hhrl_lstm_workspace_floats(B, self.T, self.o, self.H)
        return dev_empty(n, device=device)

    def fwd(self, obs: int, ldobs: int, out: int, ldo: int, ws: torch.Tensor, B: int) -> None:
        check(self._lib.hhrl_lstm_forward(obs, ldobs, self.params, out, ldo, ws.data_ptr(), B, self.T, self.o, self.H,
                                          _lib.stream_ptr()))

    def bwd(self, d_out: int, lddo: int, ws: torch.Tensor, B: int) -> None:
        check(self._lib.hhrl_lstm_backward(self.params, d_out, lddo, self.grads, ws.data_ptr(), B, self.T, self.o,
                                           self.H, _lib.stream_ptr()))


def _lstm_descs(items, forward: bool):
    """items: (BiLstm, a, lda, ws) with a = observations (forward: + out, ldo appended) or d_out."""
    n = len(items)
    arr = (_lib.LstmDesc * n)()
    for d, it in zip(arr, items):
        l = it[0]
        d.o = l.o
        d.params = C.cast(l.params, C.c_void_p)
        d.grads = C.cast(l.grads, C.c_void_p) if l.grads is not None else None
        if forward:
            _, obs, ldobs, out, ldo, ws = it[:6]
            d.obs, d.ldobs, d.out, d.ldo = obs, ldobs, out, ldo
            d.no_backward = 1 if len(it) > 6 and it[6] else 0
        else:
            _, d_out, lddo, ws = it
            d.d_out, d.lddo = d_out, lddo
        d.ws = ws.data_ptr()
    return arr


def lstm_forward_group(items, B: int) -> None:
    """items: [(BiLstm, obs_ptr, ldobs, out_ptr, ldo, workspace[, no_backward])]: LSTM modules sharing T and H,
    evaluated on B rows each, every stage one launch for the whole group.  no_backward: nobody back-propagates
    this evaluation (a target network, a feature whose gradient is not wanted): its per-step states are not saved."""
    lib = _lib.load()
    T, H = items[0][0].T, items[0][0].H
    for it in items:
        if it[0].T != T or it[0].H != H:
            raise ValueError("grouped LSTMs must share T and H")
    for i in range(0, len(items), _lib.LSTM_GROUP_MAX):
        part = items[i:i + _lib.LSTM_GROUP_MAX]
        check(lib.hhrl_lstm_forward_group(_lstm_descs(part, True), len(part), B, T, H, _lib.stream_ptr()))


def lstm_backward_group(items, B: int) -> None:
    """items: [(BiLstm, d_out_ptr, lddo, workspace)] for modules whose forward ran through their workspace."""
    lib = _lib.load()
    T, H = items[0][0].T, items[0][0].H
    for i in range(0, len(items), _lib.LSTM_GROUP_MAX):
        part = items[i:i + _lib.LSTM_GROUP_MAX]
        check(lib.hhrl_lstm_backward_group(_lstm_descs(part, False), len(part), B, T, H, _lib.stream_ptr()))


def sum_pool(inp: int, ldi: int, out: int, ldo: int, B: int, N: int, Cc: int) -> None:
    check(_lib.load().hhrl_sum_pool(inp, ldi, out, ldo, B, N, Cc, _lib.stream_ptr()))


def broadcast_rows(inp: int, ldi: int, out: int, ldo: int, B: int, N: int, Cc: int) -> None:
    check(_lib.load().hhrl_broadcast_rows(inp, ldi, out, ldo, B, N, Cc, _lib.stream_ptr()))


def copy2d(inp: int, ldi: int, out: int, ldo: int, M: int, Cc: int) -> None:
    check(_lib.load().hhrl_copy2d(inp, ldi, out, ldo, M, Cc, _lib.stream_ptr()))


class CopyQueue:
    """Collects [M, C] block copies that do not depend on each other and issues them as one launch."""

    def __init__(self, M: int, gather=None, fetch=None):
        """gather = (rows_ptr, row_width, table_ptr, n_table_rows, idx_ptr): the [M, row_width] block at rows_ptr is
        itself row idx[r] of a device-resident table.  The queue then fills that block in the same launch, and copies
        that read a column range of it read the table directly (they do not wait for the gathered rows).
        fetch = ((pinned_ptr, device_ptr, n_floats), ...): the launch also copies these pieces of a staging block from
        pinned host memory into device memory, and whatever the queue reads out of their device copy (sources, row
        numbers) it reads from the pinned block instead -- the device copy is being written by this very launch."""
        self.M = M
        self.items = []
        self.gather = gather
        self.fetch = fetch
        for src, dst, n in (fetch or ()):
            self.items.append((src, 0, dst, 0, n, 2))
        if gather is not None:
            rows, W, table, n_in, idx = gather
            self.items.append((table, W, rows, W, W, 1))

    def _pinned(self, p: int) -> int:
        """Address p, or its counterpart in the pinned block when p lies in the block's device copy."""
        for src, dst, n in (self.fetch or ()):
            if dst <= p < dst + 4 * n:
                return p - dst + src
        return p

    def add(self, inp: int, ldi: int, out: int, ldo: int, Cc: int) -> None:
        if self.gather is not None:
            rows, W, table, n_in, idx = self.gather
            off = inp - rows
            if ldi == W and 0 <= off and off + 4 * Cc <= 4 * W:      # a column range of the gathered rows
                self.items.append((table + off, W, out, ldo, Cc, 1))
                return
        self.items.append((self._pinned(inp), ldi, out, ldo, Cc, 0))

    def add_normals(self, out: int, n: int, seed_offset: int) -> None:
        """n standard normals at `out`: elements offset .. offset + n - 1 of the stream of `seed`, the two 64-bit
        integers at address `seed_offset` (read by the launch itself: a replay draws what the caller has put there)."""
        self.items.append((seed_offset, 0, out, 0, n, 3))

    def add_scatter(self, src: int, table: int, ld: int, Cc: int) -> None:
        """Rows on their way into a device-resident table [n_table_rows][ld] (gather mode): `src` points at
        {int32 count <= 63; int32 position[63]} and, 64 floats on, the rows [count][Cc]; read by the launch itself."""
        self.items.append((src, 0, table, ld, Cc, 4))

    def add_zero(self, out: int, n: int) -> None:
        """n floats at `out` set to zero in the same launch (a gradient buffer's zero_grad)."""
        self.items.append((None, 0, out, 0, n, 0))

    def flush(self) -> None:
        lib = _lib.load()
        for i in range(0, len(self.items), _lib.COPY_GROUP_MAX):
            part = self.items[i:i + _lib.COPY_GROUP_MAX]
            if self.gather is not None or self.fetch is not None:
                arr = (_lib.GCopyDesc * len(part))()
                for d, (inp, ldi, out, ldo, Cc, ga) in zip(arr, part):
                    d.inp, d.ldi, d.out, d.ldo, d.C, d.gather = inp, ldi, out, ldo, Cc, ga
                idx, n_in = (self._pinned(self.gather[4]), self.gather[3]) if self.gather is not None else (None, 0)
                check(lib.hhrl_gather_copy_group(arr, len(part), self.M, idx, n_in, _lib.stream_ptr()))
                continue
            arr = (_lib.CopyDesc * len(part))()
            for d, (inp, ldi, out, ldo, Cc, _) in zip(arr, part):
                d.inp, d.ldi, d.out, d.ldo, d.C = inp, ldi, out, ldo, Cc
            check(lib.hhrl_copy2d_group(arr, len(part), self.M, _lib.stream_ptr()))
        self.items = []


def fill(p: int, n: int, value: float) -> None:
    check(_lib.load().hhrl_fill(p, n, value, _lib.stream_ptr()))


def polyak(target: FlatBuffer, net: FlatBuffer, tau: float) -> None:
    check(_lib.load().hhrl_polyak_update(target.base_ptr, net.base_ptr, net.layout.size, tau, _lib.stream_ptr()))
