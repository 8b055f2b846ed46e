"""DeepSets VAE trainer on MI355X — the reference's `Encoder`
(algorithm.py:76-138) with the same call surface: learn / encode / decode /
save / load, `goal_dim`, and `state_dict` key names.

Every arithmetic step runs in hand-written HIP kernels (libhhrl_hip.so);
torch is used for device memory and host<->device copies only.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib, ops
from ._lib import ACT_NONE, ACT_RELU, ACT_SIGMOID, check
from .flat import FlatBuffer, Layout
from .ops import Lin, P
from .optim import FlatAdam

ENC_KEYS = ["phi.0", "phi.2", "rho.0", "rho.2", "to_mean", "to_logvar"]
DEC_KEYS = ["decoder.0", "decoder.2", "loc", "func.0", "func.2", "func.4"]


def vae_shapes(goal_dim: int, car_info_dim: int = 2) -> List[Tuple[str, Tuple[int, ...]]]:
    """Parameter names/shapes in `Encoder.parameters()` order.  Layer widths:
    DeepSetEncoder(rho_dim=64, phi_dim=32, hidden_dim=128) at algorithm.py:80,
    DeepSetDecoder(hidden_dim=128) at algorithm.py:51-66."""
    dims = {
        "encoder.phi.0": (16, car_info_dim), "encoder.phi.2": (32, 16),
        "encoder.rho.0": (128, 32), "encoder.rho.2": (64, 128),
        "encoder.to_mean": (goal_dim, 64), "encoder.to_logvar": (goal_dim, 64),
        "decoder.decoder.0": (128, goal_dim), "decoder.decoder.2": (32, 128),
        "decoder.loc": (32, 1),
        "decoder.func.0": (128, 64), "decoder.func.2": (32, 128), "decoder.func.4": (1, 32),
    }
    out = []
    for k in ["encoder." + e for e in ENC_KEYS] + ["decoder." + d for d in DEC_KEYS]:
        out.append((k + ".weight", dims[k]))
        out.append((k + ".bias", (dims[k][0],)))
    return out


def _default_init(layout: Layout, gen: torch.Generator) -> Dict[str, torch.Tensor]:
    """nn.Linear's default: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and bias."""
    out = {}
    fan_in = 1
    for name, shp in layout.shapes.items():
        if name.endswith(".weight"):
            fan_in = shp[1]
        k = 1.0 / (fan_in ** 0.5)
        out[name] = (torch.rand(shp, generator=gen) * 2 - 1) * k
    return out


class _Work:
    """Activation / gradient buffers for one (B, N) batch shape."""

    def __init__(self, B: int, N: int, g: int, dev):
        M = B * N
        e = lambda *s: ops.dev_empty(*s, device=dev)
        self.B, self.N, self.M = B, N, M
        self.h1, self.h2 = e(M, 16), e(M, 32)
        self.pooled, self.r1, self.r2 = e(B, 32), e(B, 128), e(B, 64)
        self.ml, self.z = e(B, 2 * g), e(B, g)
        self.d1, self.d2 = e(B, 128), e(B, 32)
        self.cat, self.f1, self.f2, self.dec = e(M, 64), e(M, 128), e(M, 32), e(M, 1)
        # gradients
        self.d_dec, self.d_f2, self.d_f1, self.d_cat = e(M, 1), e(M, 32), e(M, 128), e(M, 64)
        self.d_d2, self.d_d1, self.d_z = e(B, 32), e(B, 128), e(B, g)
        self.d_ml, self.d_r2, self.d_r1, self.d_pooled = e(B, 2 * g), e(B, 64), e(B, 128), e(B, 32)
        self.d_h2, self.d_h1 = e(M, 32), e(M, 16)


class _InferIO:
    """One pinned host block each way for the inference calls: a call writes all of its inputs into the host
    block, launches, and waits for the stream.  Small blocks (every per-lane, per-intersection call of the agents)
    are read and written by the kernels in place (`_lib.zero_copy_ok`): one launch per call; larger ones are staged
    through device blocks with ONE transfer each way.  Blocks grow by doubling; a call owns them until it returns
    (the class surface is used from one thread, like the reference's)."""

    def __init__(self, dev):
        self.dev = dev
        self.n_in = self.n_out = 0
        self.h_in = self.d_in = self.h_out = self.d_out = None
        self.zc_in = self.zc_out = False

    def reserve(self, n_in: int, n_out: int) -> None:
        if n_in > self.n_in:
            self.n_in = max(1024, 2 * self.n_in, n_in)
            self.h_in = torch.zeros(self.n_in, dtype=torch.float32).pin_memory()
            self.zc_in = _lib.zero_copy_ok(self.h_in)
            self.d_in = self.h_in if self.zc_in else ops.dev_empty(self.n_in, device=self.dev)
            self.h_in_np = self.h_in.numpy()
            self.h_in_i32 = self.h_in_np.view(np.int32)
        if n_out > self.n_out:
            self.n_out = max(1024, 2 * self.n_out, n_out)
            self.h_out = torch.zeros(self.n_out, dtype=torch.float32).pin_memory()
            self.zc_out = _lib.zero_copy_ok(self.h_out)
            self.d_out = self.h_out if self.zc_out else ops.dev_empty(self.n_out, device=self.dev)
            self.h_out_np = self.h_out.numpy()

    def upload(self, n: int) -> None:
        if not self.zc_in:
            self.d_in[:n].copy_(self.h_in[:n], non_blocking=True)

    def download(self, n: int) -> np.ndarray:
        if not self.zc_out:
            self.h_out[:n].copy_(self.d_out[:n], non_blocking=True)
        _lib.current_stream().synchronize()
        return self.h_out_np[:n].copy()


class Encoder:
    """Drop-in for the reference's `Encoder(car_info_dim, goal_dim)`.

    Extras that the reference does not have (all optional):
      * `learn(data, eps=...)` takes the reparameterisation noise explicitly
        (the reference draws it with `randn_like`, algorithm.py:89);
      * `learn(..., sync=False)` returns the loss as a 0-dim device tensor
        instead of forcing a device->host sync per step (algorithm.py:107);
      * `data` may already be a device tensor [B, N, 2];
      * `group` / `world_size`: data-parallel training, one flat gradient
        all-reduce per step.
    """

    LOSS_RING = 256
    EPS_POOL = 1 << 20

    def __init__(self, car_info_dim: int, goal_dim: int, device: Optional[str] = None, seed: Optional[int] = None,
                 group=None, fused: bool = True, static_shards: bool = False):
        if car_info_dim != 2:
            raise ValueError("the lane state is (location, speed): car_info_dim must be 2 (algorithm.py:94-95)")
        if not torch.cuda.is_available():
            raise RuntimeError("hhrl_vae_amd needs an AMD GPU (no CPU fallback)")
        self._lib = _lib.load()
        self.goal_dim = int(goal_dim)
        self.device = torch.device(device or "cuda")
        self.layout = Layout(vae_shapes(self.goal_dim))
        self.params = FlatBuffer(self.layout, self.device)
        gen = torch.Generator().manual_seed(3407 if seed is None else seed)
        self.params.load(_default_init(self.layout, gen))
        self.opt = FlatAdam(self.params, lr=0.001)                      # algorithm.py:83
        g = self.opt.g
        L = lambda name, act=ACT_NONE: Lin(self.params, g, name, act)
        self.phi0, self.phi2 = L("encoder.phi.0", ACT_RELU), L("encoder.phi.2")
        self.rho0, self.rho2 = L("encoder.rho.0", ACT_RELU), L("encoder.rho.2")
        self.to_mean, self.to_logvar = L("encoder.to_mean"), L("encoder.to_logvar")
        self.dec0, self.dec2 = L("decoder.decoder.0", ACT_RELU), L("decoder.decoder.2")
        self.loc = L("decoder.loc")
        self.func0, self.func2 = L("decoder.func.0", ACT_RELU), L("decoder.func.2", ACT_RELU)
        self.func4 = L("decoder.func.4", ACT_SIGMOID)
        self._work: Dict[Tuple[int, int], _Work] = {}
        # fused = True: the four-launch fused training step; False: one launch per layer
        # (same arithmetic, kept as a cross-check and for latent sizes the fused kernels do not cover)
        self.fused = bool(fused) and self.goal_dim in (2, 4, 8, 16)
        self._fused_ws: Dict[Tuple[int, int], torch.Tensor] = {}
        import ctypes
        self._offsets = (ctypes.c_long * 24)(*[self.layout.offsets[n] for n in self.layout.names])
        # losses land in a ring of device scalars: learn(sync=False) hands out a view of the
        # step's slot, valid for the next LOSS_RING - 1 steps, instead of copying it
        self._loss_ring = ops.dev_zeros(self.LOSS_RING, device=self.device)
        self._loss_slot = 0
        # views of every slot made once: a step then pays no tensor construction for its loss
        self._loss_views = [self._loss_ring[i:i + 1] for i in range(self.LOSS_RING)]
        self._loss_scalars = [self._loss_ring[i] for i in range(self.LOSS_RING)]
        self._loss_ptrs = [v.data_ptr() for v in self._loss_views]
        self._loss = self._loss_views[0]
        self.group = group
        self.world_size, self.rank = 1, 0
        if group is not None:
            import torch.distributed as dist
            self.world_size, self.rank = dist.get_world_size(group), dist.get_rank(group)
        # Noise generators (reparameterisation, algorithm.py:89, 125).  Data-parallel ranks share `seed` (identical
        # initial weights) but hold different lanes: their noise streams are offset by the rank, so the global
        # batch carries independent draws.
        noise_seed = (3407 if seed is None else seed) + 1000003 * self.rank
        self._noise_seed, self._noise_off = int(noise_seed), 0          # position in the device generator's stream
        self._eps_pool = None           # reparameterisation noise is drawn EPS_POOL floats at a time
        self._eps_used = 0
        # static_shards: every rank hands the same shard shape to every learn() call, so the global element
        # count is all-reduced once per shard shape and reused (no host round trip per step)
        self.static_shards = bool(static_shards)
        self._global_counts: Dict[int, int] = {}
        self._io = _InferIO(self.device)
        self._host_rng = np.random.default_rng(noise_seed)      # decode()'s noise (algorithm.py:125)

    # ------------------------------------------------------------ nn.Module manners the callers use
    def to(self, device=None, *args, **kwargs) -> "Encoder":
        """`Encoder(2, goal_dim).to(device)` (VAE_train.py:64; the reference's Encoder is an nn.Module).
        Parameters live on the GPU chosen at construction: a matching (or unspecified) CUDA device is
        accepted and `self` returned; anything else raises, there is no CPU path."""
        if device is not None and not isinstance(device, (torch.dtype,)):
            d = torch.device(device)
            if d.type != "cuda" or (d.index is not None and self.device.index is not None and d.index != self.device.index):
                raise RuntimeError("Encoder lives on %s and cannot move to %s (no CPU path; construct it with device=...)"
                                   % (self.device, d))
        return self

    def cuda(self, device=None) -> "Encoder":
        return self.to("cuda" if device is None else device)

    def train(self, mode: bool = True) -> "Encoder":
        """No dropout or batch statistics anywhere in the model: training and evaluation modes coincide."""
        return self

    def eval(self) -> "Encoder":
        return self

    def parameters(self):
        """Views of the flat parameter buffer in `nn.Module.parameters()` order (algorithm.py:83)."""
        return [self.params[n] for n in self.layout.names]

    # ------------------------------------------------------------ helpers
    def _ws(self, B: int, N: int) -> _Work:
        w = self._work.get((B, N))
        if w is None:
            if len(self._work) > 64:          # ragged bucket sizes: keep the cache bounded
                self._work.clear()
            w = self._work[(B, N)] = _Work(B, N, self.goal_dim, self.device)
        return w

    def _to_device(self, data) -> torch.Tensor:
        if isinstance(data, torch.Tensor):
            t = data.to(device=self.device, dtype=torch.float32)
        else:
            t = torch.as_tensor(np.asarray(data, dtype=np.float32)).to(self.device)     # algorithm.py:93
        return t.contiguous()

    # ------------------------------------------------------------ forward
    def _encode_fwd(self, x: torch.Tensor, w: _Work) -> None:
        """DeepSetEncoder.forward (algorithm.py:40-47) on [B*N, 2] rows -> w.ml = mean||logvar."""
        B, N, M, g = w.B, w.N, w.M, self.goal_dim
        self.phi0.fwd(P(x), 2, P(w.h1), 16, M)
        self.phi2.fwd(P(w.h1), 16, P(w.h2), 32, M)
        ops.sum_pool(P(w.h2), 32, P(w.pooled), 32, B, N, 32)
        self.rho0.fwd(P(w.pooled), 32, P(w.r1), 128, B)
        self.rho2.fwd(P(w.r1), 128, P(w.r2), 64, B)
        self.to_mean.fwd(P(w.r2), 64, P(w.ml), 2 * g, B)
        self.to_logvar.fwd(P(w.r2), 64, P(w.ml, g), 2 * g, B)

    def _decode_fwd(self, x_loc: int, ld_loc: int, w: _Work) -> None:
        """DeepSetDecoder.forward (algorithm.py:68-73).  The latent branch is
        evaluated once per lane and broadcast to its N cars, which equals the
        reference's repeat-then-MLP (algorithm.py:101) value for value."""
        B, N, M, g = w.B, w.N, w.M, self.goal_dim
        self.dec0.fwd(P(w.z), g, P(w.d1), 128, B)
        self.dec2.fwd(P(w.d1), 128, P(w.d2), 32, B)
        ops.broadcast_rows(P(w.d2), 32, P(w.cat), 64, B, N, 32)
        self.loc.fwd(x_loc, ld_loc, P(w.cat, 32), 64, M)
        self.func0.fwd(P(w.cat), 64, P(w.f1), 128, M)
        self.func2.fwd(P(w.f1), 128, P(w.f2), 32, M)
        self.func4.fwd(P(w.f2), 32, P(w.dec), 1, M)

    def _randn(self, n: int) -> torch.Tensor:
        """n standard normals from the library's counter-based device generator (hhrl_randn_fill): the next n
        elements of this encoder's stream (seed, rank)."""
        out = torch.empty(n, device=self.device, dtype=torch.float32)
        check(self._lib.hhrl_randn_fill(P(out), n, self._noise_seed, self._noise_off, _lib.stream_ptr()))
        self._noise_off += (n + 3) & ~3
        return out

    def _draw_eps(self, B: int, g: int) -> torch.Tensor:
        """[B, g] standard normals (algorithm.py:89) cut from a pool that one
        generator launch refills, so a training step pays no launch for them."""
        n = B * g
        if n > self.EPS_POOL // 4:
            return self._randn(n).view(B, g)
        if self._eps_pool is None or self._eps_used + n > self.EPS_POOL:
            # a fresh tensor each time: slices handed out earlier may still be queued on the stream
            self._eps_pool = self._randn(self.EPS_POOL)
            self._eps_used = 0
            self._eps_base = self._eps_pool.data_ptr()
        e = self._eps_pool[self._eps_used:self._eps_used + n].view(B, g)
        self._eps_used += (n + 3) & ~3          # keep slices 16-byte aligned
        return e

    def _draw_eps_ptr(self, B: int, g: int):
        """As _draw_eps for the fused step: (device address of the [B, g] block, the tensor that owns it),
        without building a view per step."""
        n = B * g
        if n > self.EPS_POOL // 4:
            e = self._randn(n).view(B, g)
            return e.data_ptr(), e
        if self._eps_pool is None or self._eps_used + n > self.EPS_POOL:
            self._eps_pool = self._randn(self.EPS_POOL)
            self._eps_used = 0
            self._eps_base = self._eps_pool.data_ptr()
        ptr = self._eps_base + 4 * self._eps_used
        self._eps_used += (n + 3) & ~3
        return ptr, self._eps_pool

    def _next_loss_slot(self) -> torch.Tensor:
        self._loss_slot = (self._loss_slot + 1) % self.LOSS_RING
        self._loss = self._loss_views[self._loss_slot]
        return self._loss

    # ------------------------------------------------------------ training
    def loss_and_grads(self, data, eps=None, recon_count: Optional[int] = None, _adam: bool = False,
                       _loss_to_tail: bool = False) -> torch.Tensor:
        """Forward + ELBO + backward into the flat gradient buffer (zeroed
        first).  Returns the loss as a 0-dim device tensor (no sync).
        recon_count: divisor of the squared-error sum (default B*N; the global
        element count on a data-parallel shard)."""
        x = self._to_device(data)
        if x.dim() != 3 or x.shape[-1] != 2 or x.shape[0] == 0 or x.shape[1] == 0:
            raise ValueError("data must be [B, N, 2] with B, N >= 1, got %s" % (tuple(x.shape),))
        B, N = int(x.shape[0]), int(x.shape[1])
        g = self.goal_dim
        eps_ptr = 0
        if eps is None and self.fused:
            eps_ptr, eps = self._draw_eps_ptr(B, g)
        elif eps is None:
            eps = self._draw_eps(B, g)
        else:
            eps = torch.as_tensor(eps, dtype=torch.float32).to(self.device).contiguous()
            if tuple(eps.shape) != (B, g):
                raise ValueError("eps must be [%d, %d], got %s" % (B, g, tuple(eps.shape)))
        st = _lib.stream_ptr()
        lib = self._lib
        if self.fused:
            ws = self._fused_ws.get((B, N))
            if ws is None:
                if len(self._fused_ws) > 64:
                    self._fused_ws.clear()
                n = lib.hhrl_vae_workspace_floats(B, N, g, self.layout.size)
                ws = self._fused_ws[(B, N)] = ops.dev_zeros(n, device=self.device)
            scale = 1.0 / float(recon_count if recon_count is not None else B * N)
            self._next_loss_slot()
            if not eps_ptr:
                eps_ptr = P(eps)
            # data-parallel: the loss lands right behind the gradient so that one all-reduce carries both
            loss_ptr = self.opt.g.tail_ptr if _loss_to_tail else self._loss_ptrs[self._loss_slot]
            if _adam:
                # forward, backward and the optimizer step in one call (single device)
                o = self.opt
                o.step_count += 1
                check(lib.hhrl_vae_learn_step(self.params.base_ptr, o.g.base_ptr, o.m.base_ptr, o.v.base_ptr,
                                              self._offsets, self.layout.size, P(x), eps_ptr, B, N, g, scale,
                                              o.lr, o.beta1, o.beta2, o.eps, o.step_count, loss_ptr, P(ws), st))
            else:
                check(lib.hhrl_vae_train_step(self.params.base_ptr, self.opt.g.base_ptr, self._offsets,
                                              self.layout.size, P(x), eps_ptr, B, N, g, scale, loss_ptr, P(ws), st))
            self._keep = (x, eps)
            return self._loss_scalars[self._loss_slot]
        w = self._ws(B, N)
        M = w.M
        self.opt.zero_grad()                                            # algorithm.py:98
        self._next_loss_slot().zero_()
        self._encode_fwd(x, w)                                          # :99
        check(lib.hhrl_vae_reparam(P(w.ml), P(eps), P(w.z), B, g, st))  # :100
        self._decode_fwd(P(x), 2, w)                                    # :101
        scale = 1.0 / float(recon_count if recon_count is not None else M)
        check(lib.hhrl_vae_elbo(P(w.dec), P(x), P(w.ml), scale, P(self._loss), P(w.d_dec), B, N, g, st))  # :102-104
        # backward (:105)
        self.func4.bwd(P(w.f2), 32, P(w.dec), 1, P(w.d_dec), 1, P(w.d_f2), 32, M)
        self.func2.bwd(P(w.f1), 128, P(w.f2), 32, P(w.d_f2), 32, P(w.d_f1), 128, M)
        self.func0.bwd(P(w.cat), 64, P(w.f1), 128, P(w.d_f1), 128, P(w.d_cat), 64, M)
        self.loc.bwd(P(x), 2, None, 0, P(w.d_cat, 32), 64, None, 0, M)
        ops.sum_pool(P(w.d_cat), 64, P(w.d_d2), 32, B, N, 32)
        self.dec2.bwd(P(w.d1), 128, None, 0, P(w.d_d2), 32, P(w.d_d1), 128, B)
        self.dec0.bwd(P(w.z), g, P(w.d1), 128, P(w.d_d1), 128, P(w.d_z), g, B)
        check(lib.hhrl_vae_latent_grad(P(w.d_z), P(w.ml), P(eps), P(w.d_ml), B, g, st))
        self.to_mean.bwd(P(w.r2), 64, None, 0, P(w.d_ml), 2 * g, P(w.d_r2), 64, B)
        self.to_logvar.bwd(P(w.r2), 64, None, 0, P(w.d_ml, g), 2 * g, P(w.d_r2), 64, B, dx_accumulate=1)
        self.rho2.bwd(P(w.r1), 128, None, 0, P(w.d_r2), 64, P(w.d_r1), 128, B)
        self.rho0.bwd(P(w.pooled), 32, P(w.r1), 128, P(w.d_r1), 128, P(w.d_pooled), 32, B)
        ops.broadcast_rows(P(w.d_pooled), 32, P(w.d_h2), 32, B, N, 32)
        self.phi2.bwd(P(w.h1), 16, None, 0, P(w.d_h2), 32, P(w.d_h1), 16, M)
        self.phi0.bwd(P(x), 2, P(w.h1), 16, P(w.d_h1), 16, None, 0, M)
        self._keep = (x, eps)           # keep inputs alive until the stream has consumed them
        return self._loss[0]

    def learn(self, data, eps=None, sync: bool = True, global_count: Optional[int] = None):
        """Encoder.learn (algorithm.py:92-107): one Adam step on one batch of
        lanes that all hold N cars.  Returns the loss (float, or a device
        scalar when sync=False).

        Data-parallel (group set): `data` is this rank's shard; the squared
        error is divided by the GLOBAL element count (`global_count`, or an
        all-reduced count when omitted: a host round trip per step unless the
        encoder was built with static_shards=True), KL stays a sum, and the flat gradient
        buffer is all-reduced (SUM) once before the Adam step, so every rank
        applies the single-device update for the global batch."""
        if self.group is None:
            loss = self.loss_and_grads(data, eps, _adam=self.fused)
            if not self.fused:
                self.opt.step()                                         # :106
        else:
            from . import dist as hdist
            x = self._to_device(data)
            if global_count is None:
                local = int(x.shape[0] * x.shape[1])
                global_count = self._global_counts.get(local) if self.static_shards else None
                if global_count is None:
                    global_count = hdist.global_count(local, self.group, self.device)     # one small all-reduce + host read
                    if self.static_shards:
                        self._global_counts[local] = global_count
            g = self.opt.g
            if self.fused:
                loss = self.loss_and_grads(x, eps, recon_count=global_count, _loss_to_tail=True)
                hdist.allreduce_sum_(g.full, self.group)                # one flat bucket: gradient + loss, SUM
                loss = self._loss_scalars[self._loss_slot]
                self.opt.step(copy=(g.tail_ptr, self._loss_ptrs[self._loss_slot]))  # :106, and the summed loss into its slot
            else:
                loss = self.loss_and_grads(x, eps, recon_count=global_count)
                hdist.allreduce_sum_(g.data, self.group)
                loss = hdist.allreduce_sum_(loss.clone(), self.group)
                self.opt.step()                                         # :106
        return float(loss.item()) if sync else loss

    def learn_many(self, chunks, eps_list=None):
        """`learn(chunk, sync=False)` for every chunk of a list, in order, with ONE library call per group of
        chunks (hhrl_vae_learn_steps): the loop of VAE_train.py:65-68, whose 64-lane steps take less GPU time than
        a Python call costs.  chunks: fp32 CUDA tensors [B, N, 2] (vae_train.prepare_chunks); eps_list: optional
        noise per chunk.  Returns the losses as device scalars, valid until LOSS_RING further steps have run.
        Single device, fused path only; anything else falls back to one learn() per chunk."""
        import ctypes
        chunks = list(chunks)
        if self.group is not None or not self.fused or not all(
                isinstance(c, torch.Tensor) and c.is_cuda and c.dtype == torch.float32 and c.dim() == 3 and c.shape[-1] == 2
                and c.shape[0] > 0 and c.shape[1] > 0 and c.is_contiguous() for c in chunks):
            return [self.learn(c, eps=None if eps_list is None else eps_list[i], sync=False) for i, c in enumerate(chunks)]
        lib, g, o = self._lib, self.goal_dim, self.opt
        out = []
        group = self.LOSS_RING - 1
        for lo in range(0, len(chunks), group):
            part = chunks[lo:lo + group]
            n = len(part)
            xs, es, Bs, Ns = (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)(), (ctypes.c_int * n)(), (ctypes.c_int * n)()
            ls, wss = (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)()
            keep = []
            for k, x in enumerate(part):
                B, N = int(x.shape[0]), int(x.shape[1])
                if eps_list is None:
                    e_ptr, e = self._draw_eps_ptr(B, g)
                else:
                    e = torch.as_tensor(eps_list[lo + k], dtype=torch.float32).to(self.device).contiguous()
                    if tuple(e.shape) != (B, g):
                        raise ValueError("eps must be [%d, %d], got %s" % (B, g, tuple(e.shape)))
                    e_ptr = P(e)
                ws = self._fused_ws.get((B, N))
                if ws is None:
                    if len(self._fused_ws) > 64:
                        self._fused_ws.clear()
                    ws = self._fused_ws[(B, N)] = ops.dev_zeros(lib.hhrl_vae_workspace_floats(B, N, g, self.layout.size),
                                                                device=self.device)
                self._next_loss_slot()
                xs[k], es[k], Bs[k], Ns[k], ls[k], wss[k] = P(x), e_ptr, B, N, self._loss_ptrs[self._loss_slot], P(ws)
                out.append(self._loss_scalars[self._loss_slot])
                keep.append((x, e, ws))
            check(lib.hhrl_vae_learn_steps(self.params.base_ptr, o.g.base_ptr, o.m.base_ptr, o.v.base_ptr, self._offsets,
                                           self.layout.size, n, xs, es, Bs, Ns, g, o.lr, o.beta1, o.beta2, o.eps,
                                           o.step_count + 1, ls, wss, _lib.stream_ptr()))
            o.step_count += n
            self._keep = keep
        return out

    # ------------------------------------------------------------ inference
    def _encode_dev(self, cars: torch.Tensor, lane_start: Optional[torch.Tensor], B: int, N: int) -> torch.Tensor:
        out = ops.dev_empty(B, 2 * self.goal_dim, device=self.device)
        if self.fused:
            check(self._lib.hhrl_vae_encode(self.params.base_ptr, self._offsets, self.layout.size, P(cars),
                                            lane_start.data_ptr() if lane_start is not None else None, B, N,
                                            self.goal_dim, P(out), 2 * self.goal_dim, _lib.stream_ptr()))
            self._keep = (cars, lane_start)
            return out
        w = self._ws(B, N)
        self._encode_fwd(cars, w)
        return w.ml.clone()

    def encode(self, x) -> np.ndarray:
        """Encoder.encode (algorithm.py:109-116): one lane [N, 2] -> mean||logvar [2g]."""
        if len(x) == 0:
            return np.zeros((self.goal_dim * 2))
        if self.fused and not isinstance(x, torch.Tensor):
            xs = np.asarray(x, dtype=np.float32)
            if xs.ndim != 2 or xs.shape[1] != 2:
                raise ValueError("encode expects one lane [N, 2]")
            return self._encode_host(xs, None, 1, int(xs.shape[0]))[0]
        xt = self._to_device(x)
        if xt.dim() != 2 or xt.shape[1] != 2:
            raise ValueError("encode expects one lane [N, 2]")
        return self._encode_dev(xt, None, 1, int(xt.shape[0]))[0].cpu().numpy()

    def _encode_host(self, cars: np.ndarray, start: Optional[np.ndarray], B: int, N: int) -> np.ndarray:
        """Fused encoder on host data: cars [n, 2] (and the lanes' start rows, int32 [B + 1]) go up in one
        pinned block, mean||logvar [B, 2g] comes back in one."""
        io, g2 = self._io, 2 * self.goal_dim
        n_car = cars.size
        n_car_al = (n_car + 3) & ~3                          # lane_start begins on a 16-byte boundary
        n_in = n_car_al + (B + 1 if start is not None else 0)
        io.reserve(n_in, B * g2)
        io.h_in_np[:n_car] = cars.reshape(-1)
        if start is not None:
            io.h_in_i32[n_car_al:n_in] = start
        io.upload(n_in)
        check(self._lib.hhrl_vae_encode(self.params.base_ptr, self._offsets, self.layout.size, P(io.d_in),
                                        P(io.d_in, n_car_al) if start is not None else None, B, N,
                                        self.goal_dim, P(io.d_out), g2, _lib.stream_ptr()))
        return io.download(B * g2).reshape(B, g2)

    def encode_batch(self, lanes: torch.Tensor) -> torch.Tensor:
        """[B, N, 2] lanes with a common car count -> [B, 2g] on the device."""
        xt = self._to_device(lanes)
        return self._encode_dev(xt, None, int(xt.shape[0]), int(xt.shape[1]))

    def encode_lanes(self, lanes: Sequence) -> np.ndarray:
        """Several lanes of different lengths (each [n_i, 2], possibly empty) in one launch -> [len(lanes), 2g]:
        what agent.py:326, 748 build with one encode() call per lane."""
        B = len(lanes)
        if B == 0:
            return np.zeros((0, 2 * self.goal_dim), dtype=np.float32)
        counts = [len(l) for l in lanes]
        if not self.fused or sum(counts) == 0:
            return np.stack([np.asarray(self.encode(l), dtype=np.float32) for l in lanes])
        cars = np.concatenate([np.asarray(l, dtype=np.float32).reshape(-1, 2) for l in lanes if len(l)], axis=0)
        start = np.zeros(B + 1, dtype=np.int32)
        np.cumsum(counts, out=start[1:])
        return self._encode_host(cars, start, B, 0)

    def decode_batch(self, goal, loc, eps=None) -> torch.Tensor:
        """goal [B, 2g] (mean||logvar), loc [B, N] -> decoded speeds [B, N] on the device.  eps [B, g]:
        reparameterisation noise (drawn on the device when omitted, algorithm.py:125)."""
        g = self.goal_dim
        goal_t = torch.as_tensor(goal, dtype=torch.float32).reshape(-1, 2 * g).to(self.device).contiguous()
        B = int(goal_t.shape[0])
        loc_t = torch.as_tensor(loc, dtype=torch.float32).reshape(B, -1).to(self.device).contiguous()
        N = int(loc_t.shape[1])
        if eps is None:
            eps_t = self._randn(B * g).view(B, g)
        else:
            eps_t = torch.as_tensor(eps, dtype=torch.float32).reshape(B, g).to(self.device).contiguous()
        if self.fused:
            out = ops.dev_empty(B, N, device=self.device)
            check(self._lib.hhrl_vae_decode(self.params.base_ptr, self._offsets, self.layout.size, P(goal_t), 2 * g, P(eps_t),
                                            P(loc_t), N, B, N, g, P(out), N, _lib.stream_ptr()))
            self._keep = (goal_t, eps_t, loc_t)
            return out
        outs = []
        for n in range(N):                       # per-layer path: one location column at a time
            w = self._ws(B, 1)
            w.ml.copy_(goal_t)
            check(self._lib.hhrl_vae_reparam(P(w.ml), P(eps_t), P(w.z), B, g, _lib.stream_ptr()))
            col = loc_t[:, n:n + 1].contiguous()
            self._decode_fwd(P(col), 1, w)
            outs.append(w.dec.clone())
        return torch.cat(outs, dim=1)

    def decode(self, goal, loc, eps=None) -> np.ndarray:
        """Encoder.decode (algorithm.py:121-127): samples z from (mean, logvar)
        and decodes the speed at `loc`; shapes [2g] and [1] -> [1]."""
        g = self.goal_dim
        goal_np = np.asarray(goal, dtype=np.float32).reshape(1, 2 * g)
        loc_np = np.asarray(loc, dtype=np.float32).reshape(1, 1)
        eps_np = None if eps is None else np.asarray(eps, dtype=np.float32).reshape(1, g)
        if self.fused:
            return self.decode_host(goal_np, loc_np, eps_np)[0]
        return self.decode_batch(goal_np, loc_np, eps_np)[0].cpu().numpy()

    def decode_host(self, goal, loc, eps=None) -> np.ndarray:
        """decode_batch for host data with the result on the host: goal [B, 2g], loc [B, N] and the noise
        [B, g] (drawn with the host generator seeded at construction when omitted) go up in one pinned block,
        the speeds [B, N] come back in one."""
        if not self.fused:
            return self.decode_batch(goal, loc, eps).cpu().numpy()
        g = self.goal_dim
        goal_np = np.asarray(goal, dtype=np.float32).reshape(-1, 2 * g)
        B = goal_np.shape[0]
        loc_np = np.asarray(loc, dtype=np.float32).reshape(B, -1)
        N = loc_np.shape[1]
        if eps is None:
            eps_np = self._host_rng.standard_normal((B, g), dtype=np.float32)
        else:
            eps_np = np.asarray(eps, dtype=np.float32).reshape(B, g)
        io = self._io
        o_eps, o_loc = B * 2 * g, B * 3 * g
        n_in = o_loc + B * N
        io.reserve(n_in, B * N)
        io.h_in_np[:o_eps] = goal_np.reshape(-1)
        io.h_in_np[o_eps:o_loc] = eps_np.reshape(-1)
        io.h_in_np[o_loc:n_in] = loc_np.reshape(-1)
        io.upload(n_in)
        check(self._lib.hhrl_vae_decode(self.params.base_ptr, self._offsets, self.layout.size, P(io.d_in), 2 * g,
                                        P(io.d_in, o_eps), P(io.d_in, o_loc), N, B, N, g, P(io.d_out), N, _lib.stream_ptr()))
        return io.download(B * N).reshape(B, N)

    # ------------------------------------------------------------ checkpoints
    def state_dict(self) -> Dict[str, torch.Tensor]:
        return self.params.state_dict()

    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        self.params.load(sd)

    def grads(self) -> Dict[str, torch.Tensor]:
        return self.opt.g.state_dict()

    def save(self, filename: str) -> None:
        """Same three files as algorithm.py:129-132, same key names."""
        sd = self.params.state_dict()
        torch.save({k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}, filename + "_enc")
        torch.save({k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")}, filename + "_dec")
        torch.save(self.opt.state_dict(), filename + "_encoder_optimizer")

    def load(self, filename: str) -> None:
        """algorithm.py:134-138."""
        enc = torch.load(filename + "_enc", weights_only=True)
        dec = torch.load(filename + "_dec", weights_only=True)
        sd = {"encoder." + k: v for k, v in enc.items()}
        sd.update({"decoder." + k: v for k, v in dec.items()})
        self.params.load(sd)
        self.opt.load_state_dict(torch.load(filename + "_encoder_optimizer", weights_only=True))
