"""Algorithmic work of the learning path, counted from layer shapes: multiply-accumulates and
compulsory memory traffic of one VAE training step (algorithm.py:92-107) and of one TD3 update of any
family (algorithm.py:283-328, 479-542, 768-838, 1109-1168, 1316-1366).  Used to place measured times on
the chip's roofline.

"Algorithmic" means the products the update's results require, whatever an implementation executes:
  * a dense layer y = act(W x + b) costs K*N MACs per row forward, K*N for dW, and K*N for dX where an
    input gradient is needed (it is not for layers fed by data);
  * the bidirectional LSTM contributes its forward direction over T steps, 4H*(o+H) MACs each, and ONE
    step of the reverse direction on x[T-1] with a zero state, 4H*o: the module's output is
    `r_out[:, -1, :]` (algorithm.py:154), which holds the reverse state after its first step only.
    (nn.LSTM itself runs the reverse direction over all T steps; those products do not reach the output.)
  * the actor loss back-propagates through head 1 of the critic for the input gradient only: the stray
    parameter gradients autograd leaves in the critic there (algorithm.py:1355-1357) are zeroed before use
    and are not counted.
FLOPs = 2 * MACs.  Bytes: the sampled rows once (fp32) plus 28 B/parameter per Adam step (read p, g, m, v;
write p, m, v) and 12 B/parameter per soft update (read p, target; write target)."""
from __future__ import annotations

from typing import Dict

from .spec import Family, actor_shapes, critic_shapes

PEAK_FP32_MATRIX_TFLOPS = 157.3     # MI355X dense fp32 MFMA (v_mfma_f32_32x32x2_f32)
PEAK_HBM_GBS = 8000.0               # MI355X HBM3E


def _numel(shapes) -> int:
    n = 0
    for _, shp in shapes:
        k = 1
        for d in shp:
            k *= int(d)
        n += k
    return n


def lstm_macs(o: int, T: int, H: int) -> Dict[str, float]:
    """Per row: forward, and backward (BPTT: dh through W_hh per step, dW_ih and dW_hh per step, the reverse
    cell's dW_ih, the fc layer's dW and dX; no gradient with respect to the observations)."""
    fwd = T * 4.0 * H * (o + H) + 4.0 * H * o + 2.0 * H * H
    bwd = T * 4.0 * H * H + T * 4.0 * H * (o + H) + 4.0 * H * o + 2.0 * (2.0 * H * H)
    return {"fwd": fwd, "bwd": bwd}


def mlp_macs(d_in: int, h0: int, h1: int, out: int, dx_in: float = 0.0) -> Dict[str, float]:
    """Three dense layers per row.  bwd_w: weight gradients; bwd_x: input gradients of layers 3 and 2 plus
    `dx_in` input columns of layer 1 that need a gradient."""
    fwd = float(d_in * h0 + h0 * h1 + h1 * out)
    return {"fwd": fwd, "bwd_w": fwd, "bwd_x": float(h0 * h1 + h1 * out) + dx_in * h0}


def vae_step(B: int, N: int, g: int) -> Dict[str, float]:
    """One Encoder.learn on B lanes of N cars, latent g.  Forward MACs per lane: per-car layers
    (phi 2*16 + 16*32, loc 32, func 64*128 + 128*32 + 32 = 12896 in all) times N, per-lane layers
    (rho 32*128 + 128*64, heads 2*64*g, decoder g*128 + 128*32 = 16384 + 256 g) once; training = forward +
    input gradients + weight gradients = 3x."""
    macs = 12896.0 * N + 16384.0 + 256.0 * g
    params = 2 * 16 + 16 + 16 * 32 + 32 + 32 * 128 + 128 + 128 * 64 + 64 + 2 * (64 * g + g) + g * 128 + 128 + 128 * 32 + 32 \
        + 32 + 32 + 64 * 128 + 128 + 128 * 32 + 32 + 32 + 1
    return {"flops": 6.0 * macs * B, "bytes": 4.0 * B * (2.0 * N + g) + 28.0 * params, "params": float(params)}


def td3_update(fam: Family, B: int) -> Dict[str, float]:
    """One `learn()` on B rows, averaged over the `policy_freq`-call cycle (every policy_freq-th call also
    steps the actors and the targets).  Returns flops, bytes, and the per-row MAC counts of a plain call and
    of the extra work of a policy call."""
    n_act = len(fam.a_dims)
    ah0, ah1 = fam.actor_hidden
    ch0, ch1 = fam.critic_hidden
    lstm = [lstm_macs(s.o, s.T, s.s) if s.T > 1 else None for s in fam.streams]
    lstm_feat = sum(s.feat for s in fam.streams if s.T > 1)
    actor = [mlp_macs(fam.streams[k].feat + fam.goal_dim, ah0, ah1, fam.a_dims[k],
                      dx_in=fam.streams[k].feat if fam.streams[k].T > 1 else 0.0) for k in range(n_act)]
    head = mlp_macs(fam.critic_in, ch0, ch1, 1, dx_in=lstm_feat)
    lf = lambda k: lstm[k]["fwd"] if lstm[k] else 0.0
    lb = lambda k: lstm[k]["bwd"] if lstm[k] else 0.0
    n_str = len(fam.streams)
    # ---- every call: target value and the critic step
    plain = sum(lf(k) + actor[k]["fwd"] for k in range(n_act))                    # target policies
    plain += sum(lf(k) for k in range(n_str)) + 2 * head["fwd"]                   # target critic
    plain += sum(lf(k) for k in range(n_str)) + 2 * head["fwd"]                   # critic forward
    plain += 2 * (head["bwd_w"] + head["bwd_x"]) + sum(lb(k) for k in range(n_str))   # critic backward
    # ---- policy call, on top: every actor's step through head 1 of the updated critic
    extra = sum(lf(k) for k in range(n_str))                                      # the critic's features, new weights
    extra += sum(lf(k) + actor[k]["fwd"] for k in range(n_act))                   # joint action
    q1_dx = float(ch0 * ch1 + ch1) + fam.act_dim * ch0                            # head 1 down to the action columns
    for i in range(n_act):
        extra += head["fwd"] + q1_dx + actor[i]["bwd_w"] + actor[i]["bwd_x"] + lb(i)
        if i + 1 < n_act:
            extra += lf(i) + actor[i]["fwd"]                                      # later actors see the new policy
    macs_row = plain + extra / fam.policy_freq
    p_critic = _numel(critic_shapes(fam))
    p_actors = sum(_numel(actor_shapes(fam, k)) for k in range(n_act))
    by = 4.0 * B * fam.row_width + 28.0 * p_critic + (28.0 * p_actors + 12.0 * (p_actors + p_critic)) / fam.policy_freq
    return {"flops": 2.0 * macs_row * B, "bytes": by, "macs_per_row": macs_row, "macs_per_row_plain": plain,
            "macs_per_row_policy_extra": extra, "params_critic": float(p_critic), "params_actors": float(p_actors)}


def roofline(flops: float, nbytes: float, seconds: float) -> Dict[str, object]:
    """Places (flops, bytes) done in `seconds` on the roofline: the bound is the one the arithmetic intensity
    selects (ridge = fp32 matrix peak / HBM peak, about 19.7 FLOP/B)."""
    ridge = PEAK_FP32_MATRIX_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)
    intensity = flops / max(nbytes, 1.0)
    if intensity >= ridge:
        achieved = flops / seconds / 1e12
        return {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP32_MATRIX_TFLOPS, "flop_per_byte": intensity}
    achieved = nbytes / seconds / 1e9
    return {"bound": "hbm", "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": achieved / PEAK_HBM_GBS,
            "flop_per_byte": intensity}
