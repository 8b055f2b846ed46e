"""Kernel-level parity: every C-ABI entry point against plain torch fp32 on the
same inputs.  Needs a GPU."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from golden_util import assert_close

pytestmark = pytest.mark.gpu

RTOL = 1e-4          # the bar stated for this path: fp32, rtol 1e-4


def scaled_close(name, got, want, rtol=RTOL, atol_rel=2e-6):
    """rtol on each element plus an absolute floor tied to the tensor's scale
    (fp32 sums in a different order cannot meet a pure rtol near zero)."""
    want = np.asarray(want, dtype=np.float64)
    scale = float(np.max(np.abs(want))) if want.size else 0.0
    assert_close(name, got, want, rtol=rtol, atol=atol_rel * max(scale, 1e-30))


@pytest.fixture(scope="module")
def env():
    import hhrl_vae_amd._lib as L
    lib = L.load()
    dev = torch.device("cuda")
    return lib, L, dev


def _act(x, act):
    return [lambda v: v, F.relu, torch.tanh, torch.sigmoid][act](x)


@pytest.mark.parametrize("M,K,N,act", [
    (64, 17, 128, 1), (64, 128, 128, 1), (64, 128, 1, 0), (37, 393, 512, 1), (4096, 64, 128, 1),
    (28672, 128, 32, 1), (28672, 32, 1, 3), (28672, 2, 16, 1), (28672, 1, 32, 0), (4096, 4, 128, 1),
    (640, 13, 512, 0), (64, 300, 4, 2), (1, 5, 3, 2), (5, 1, 1, 0),
])
def test_linear_forward_backward(env, M, K, N, act):
    lib, L, dev = env
    g = torch.Generator().manual_seed(M * 7 + K * 3 + N)
    ldx, ldy = K + 3, N + 2                       # operands live inside wider buffers
    xw = torch.randn(M, ldx, generator=g)
    w = torch.randn(N, K, generator=g) / max(K, 1) ** 0.5
    b = torch.randn(N, generator=g)
    dyw = torch.randn(M, ldy, generator=g)
    x = xw[:, :K].clone().requires_grad_(True)
    wt, bt = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = _act(F.linear(x, wt, bt), act)
    y.backward(dyw[:, :N])
    d = lambda t: t.to(dev).contiguous()
    xd, wd, bd, dyd = d(xw), d(w), d(b), d(dyw)
    yd = torch.full((M, ldy), 7.0, device=dev)
    st = L.stream_ptr()
    L.check(lib.hhrl_linear_forward(xd.data_ptr(), ldx, wd.data_ptr(), bd.data_ptr(), yd.data_ptr(), ldy, M, K, N, act, 0, st))
    scaled_close("y", yd[:, :N].cpu().numpy(), y.detach().numpy())
    assert torch.all(yd[:, N:] == 7.0), "forward wrote outside its column slice"
    # accumulate mode: y = act(y_in + x W^T + b)
    y0 = torch.randn(M, ldy, generator=g)
    y0d = d(y0)
    L.check(lib.hhrl_linear_forward(xd.data_ptr(), ldx, wd.data_ptr(), bd.data_ptr(), y0d.data_ptr(), ldy, M, K, N, act, 1, st))
    scaled_close("y_acc", y0d[:, :N].cpu().numpy(), _act(y0[:, :N] + F.linear(xw[:, :K], w, b), act).numpy())
    # backward
    dxd = torch.full((M, ldx), 3.0, device=dev)
    dwd = torch.zeros(N, K, device=dev)
    dbd = torch.zeros(N, device=dev)
    L.check(lib.hhrl_linear_backward(xd.data_ptr(), ldx, wd.data_ptr(), yd.data_ptr(), ldy, dyd.data_ptr(), ldy,
                                     dxd.data_ptr(), ldx, 0, dwd.data_ptr(), dbd.data_ptr(), M, K, N, act, st))
    scaled_close("dx", dxd[:, :K].cpu().numpy(), x.grad.numpy())
    assert torch.all(dxd[:, K:] == 3.0)
    scaled_close("dw", dwd.cpu().numpy(), wt.grad.numpy(), atol_rel=5e-6)
    scaled_close("db", dbd.cpu().numpy(), bt.grad.numpy(), atol_rel=5e-6)
    # gradients accumulate (+=) on a second call; dx accumulates on request
    dyd2 = d(dyw)
    L.check(lib.hhrl_linear_backward(xd.data_ptr(), ldx, wd.data_ptr(), yd.data_ptr(), ldy, dyd2.data_ptr(), ldy,
                                     dxd.data_ptr(), ldx, 1, dwd.data_ptr(), dbd.data_ptr(), M, K, N, act, st))
    scaled_close("dw2", dwd.cpu().numpy(), 2 * wt.grad.numpy(), atol_rel=5e-6)
    scaled_close("dx2", dxd[:, :K].cpu().numpy(), 2 * x.grad.numpy())


def test_argument_errors(env):
    lib, L, dev = env
    t = torch.zeros(16, device=dev)
    rc = lib.hhrl_linear_forward(None, 4, t.data_ptr(), None, t.data_ptr(), 4, 2, 4, 2, 0, 0, None)
    assert rc == 1 and b"null" in lib.hhrl_last_error()
    rc = lib.hhrl_linear_forward(t.data_ptr(), 2, t.data_ptr(), None, t.data_ptr(), 4, 2, 4, 2, 0, 0, None)
    assert rc == 1 and b"leading" in lib.hhrl_last_error()
    rc = lib.hhrl_adam_step(t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), 16, 1e-3, 0.9, 0.999, 1e-8, 0, 1.0, None)
    assert rc == 1
    with pytest.raises(L.HipError):
        L.check(lib.hhrl_fill(None, 4, 0.0, None))


@pytest.mark.parametrize("n", [1, 5, 64, 30912, 1216898 + 3])
def test_adam_and_polyak(env, n):
    lib, L, dev = env
    g = torch.Generator().manual_seed(n)
    p = torch.randn(n, generator=g)
    ref_p = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref_p], lr=1e-3)
    pd = p.to(dev)
    md, vd = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    for step in range(1, 4):
        gr = torch.randn(n, generator=g) * 10.0 ** (-step)
        ref_p.grad = gr.clone()
        opt.step()
        grd = gr.to(dev)
        L.check(lib.hhrl_adam_step(pd.data_ptr(), grd.data_ptr(), md.data_ptr(), vd.data_ptr(), n, 1e-3, 0.9, 0.999,
                                   1e-8, step, 1.0, L.stream_ptr()))
        scaled_close("p%d" % step, pd.cpu().numpy(), ref_p.detach().numpy(), rtol=1e-5, atol_rel=1e-7)
    st = opt.state[ref_p]
    scaled_close("m", md.cpu().numpy(), st["exp_avg"].numpy(), rtol=1e-5)
    scaled_close("v", vd.cpu().numpy(), st["exp_avg_sq"].numpy(), rtol=1e-5)
    # unaligned base pointer takes the scalar path
    if n > 8:
        q = torch.randn(n, generator=g)
        qd, gq = q.to(dev), torch.randn(n, generator=g)
        gqd, mq, vq = gq.to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        L.check(lib.hhrl_adam_step(qd.data_ptr() + 4, gqd.data_ptr() + 4, mq.data_ptr() + 4, vq.data_ptr() + 4, n - 1, 1e-3,
                                   0.9, 0.999, 1e-8, 1, 0.5, L.stream_ptr()))
        rq = q[1:].clone().requires_grad_(True)
        o2 = torch.optim.Adam([rq], lr=1e-3)
        rq.grad = 0.5 * gq[1:]
        o2.step()
        scaled_close("p_unaligned", qd[1:].cpu().numpy(), rq.detach().numpy(), rtol=1e-5, atol_rel=1e-7)
        assert float(qd[0]) == float(q[0])
    t = torch.randn(n, generator=g)
    td = t.to(dev)
    L.check(lib.hhrl_polyak_update(td.data_ptr(), pd.data_ptr(), n, 0.005, L.stream_ptr()))
    scaled_close("polyak", td.cpu().numpy(), (0.005 * pd.cpu() + (1 - 0.005) * t).numpy(), rtol=1e-6)


def test_data_movement(env):
    lib, L, dev = env
    g = torch.Generator().manual_seed(3)
    B, N, Cc, ld = 5, 7, 32, 40
    x = torch.randn(B * N, ld, generator=g)
    xd = x.to(dev)
    out = torch.zeros(B, Cc + 1, device=dev)
    L.check(lib.hhrl_sum_pool(xd.data_ptr(), ld, out.data_ptr(), Cc + 1, B, N, Cc, L.stream_ptr()))
    scaled_close("sum_pool", out[:, :Cc].cpu().numpy(), x[:, :Cc].view(B, N, Cc).sum(1).numpy())
    src = torch.randn(B, Cc, generator=g)
    dst = torch.zeros(B * N, ld, device=dev)
    srcd = src.to(dev)
    L.check(lib.hhrl_broadcast_rows(srcd.data_ptr(), Cc, dst.data_ptr() + 4 * 8, ld, B, N, Cc, L.stream_ptr()))
    assert torch.equal(dst[:, 8:8 + Cc].cpu(), src.repeat_interleave(N, 0))
    assert torch.all(dst[:, :8] == 0)
    c = torch.zeros(B * N, 12, device=dev)
    L.check(lib.hhrl_copy2d(xd.data_ptr() + 4 * 3, ld, c.data_ptr() + 4 * 2, 12, B * N, 9, L.stream_ptr()))
    assert torch.equal(c[:, 2:11].cpu(), x[:, 3:12])
    L.check(lib.hhrl_fill(c.data_ptr(), 12, -0.25, L.stream_ptr()))
    assert torch.all(c[0] == -0.25) and torch.equal(c[1, 2:11].cpu(), x[1, 3:12])


def test_vae_elementwise(env):
    lib, L, dev = env
    g = torch.Generator().manual_seed(11)
    B, N, gd = 33, 7, 4
    ml = torch.randn(B, 2 * gd, generator=g).requires_grad_(True)
    eps = torch.randn(B, gd, generator=g)
    data = torch.rand(B * N, 2, generator=g)
    dec = torch.rand(B * N, 1, generator=g).requires_grad_(True)
    mean, lv = ml[:, :gd], ml[:, gd:]
    z = mean + eps * torch.exp(0.5 * lv)
    zd = torch.zeros(B, gd, device=dev)
    st = L.stream_ptr()
    mld, epsd = ml.detach().to(dev), eps.to(dev)
    L.check(lib.hhrl_vae_reparam(mld.data_ptr(), epsd.data_ptr(), zd.data_ptr(), B, gd, st))
    scaled_close("z", zd.cpu().numpy(), z.detach().numpy())
    recon = F.mse_loss(dec, data[:, 1:])
    kl = -0.5 * torch.sum(1 + lv - mean.pow(2) - lv.exp())
    dz = torch.randn(B, gd, generator=g)
    (recon + kl + (z * dz).sum()).backward()
    loss = torch.zeros(1, device=dev)
    ddec = torch.zeros(B * N, 1, device=dev)
    decd, datad = dec.detach().to(dev), data.to(dev)          # keep device operands referenced while kernels run
    L.check(lib.hhrl_vae_elbo(decd.data_ptr(), datad.data_ptr(), mld.data_ptr(), 1.0 / (B * N),
                              loss.data_ptr(), ddec.data_ptr(), B, N, gd, st))
    scaled_close("loss", loss.cpu().numpy(), [float(recon + kl)])
    scaled_close("ddec", ddec.cpu().numpy(), dec.grad.numpy())
    dml = torch.zeros(B, 2 * gd, device=dev)
    dzd = dz.to(dev)
    L.check(lib.hhrl_vae_latent_grad(dzd.data_ptr(), mld.data_ptr(), epsd.data_ptr(), dml.data_ptr(), B, gd, st))
    scaled_close("dml", dml.cpu().numpy(), ml.grad.numpy())


def test_td3_elementwise(env):
    lib, L, dev = env
    g = torch.Generator().manual_seed(12)
    B, a = 70, 4
    pi = torch.tanh(torch.randn(B, a + 2, generator=g))
    eps = torch.randn(B, a, generator=g) * 3
    out = torch.zeros(B, a + 5, device=dev)
    st = L.stream_ptr()
    pid, epsd = pi.to(dev), eps.to(dev)
    L.check(lib.hhrl_td3_smooth_action(pid.data_ptr(), a + 2, epsd.data_ptr(), out.data_ptr() + 4 * 5, a + 5, B, a, st))
    want = (pi[:, :a] + (eps * 0.2).clamp(-0.5, 0.5)).clamp(-1, 1)
    scaled_close("smooth", out[:, 5:].cpu().numpy(), want.numpy(), rtol=1e-6)
    q1 = torch.randn(B, 1, generator=g).requires_grad_(True)
    q2 = torch.randn(B, 1, generator=g).requires_grad_(True)
    r = torch.randn(B, 1, generator=g)
    tq1, tq2 = torch.randn(B, 1, generator=g), torch.randn(B, 1, generator=g)
    y = r + 0.95 * torch.min(tq1, tq2)
    yd = torch.zeros(B, 1, device=dev)
    tq1d, tq2d, rd = tq1.to(dev), tq2.to(dev), r.to(dev)
    L.check(lib.hhrl_td3_target_q(tq1d.data_ptr(), tq2d.data_ptr(), rd.data_ptr(), 1, 0.95, yd.data_ptr(), B, st))
    scaled_close("y", yd.cpu().numpy(), y.numpy(), rtol=1e-6)
    l1, l2 = F.mse_loss(q1, y), F.mse_loss(q2, y)
    (l1 + l2).backward()
    loss = torch.zeros(2, device=dev)
    dq1, dq2 = torch.zeros(B, 1, device=dev), torch.zeros(B, 1, device=dev)
    q1d, q2d = q1.detach().to(dev), q2.detach().to(dev)
    L.check(lib.hhrl_twin_mse(q1d.data_ptr(), q2d.data_ptr(), yd.data_ptr(), 1.0 / B,
                              loss.data_ptr(), dq1.data_ptr(), dq2.data_ptr(), B, st))
    scaled_close("mse", loss.cpu().numpy(), [float(l1), float(l2)])
    scaled_close("dq1", dq1.cpu().numpy(), q1.grad.numpy())
    scaled_close("dq2", dq2.cpu().numpy(), q2.grad.numpy())


@pytest.mark.parametrize("B,T,o,H", [(9, 5, 7, 12), (64, 10, 13, 128), (64, 10, 25, 128), (3, 1, 4, 8), (130, 4, 5, 8)])
def test_lstm_forward_backward(env, B, T, o, H):
    """hhrl_lstm_forward/backward against torch.nn.LSTM(bidirectional) + Linear,
    taking r_out[:, -1, :] as the reference does (algorithm.py:152-156)."""
    lib, L, dev = env
    torch.manual_seed(B + T + o + H)
    ref = torch.nn.LSTM(o, H, 1, batch_first=True, bidirectional=True)
    fc = torch.nn.Linear(2 * H, H)
    ld = T * o + 3
    obs = torch.randn(B, ld)
    out = fc(ref(obs[:, :T * o].reshape(B, T, o))[0][:, -1, :])
    dout = torch.randn(B, H + 2)
    out.backward(dout[:, :H])
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0", "weight_ih_l0_reverse", "weight_hh_l0_reverse",
             "bias_ih_l0_reverse", "bias_hh_l0_reverse"]
    ps = [getattr(ref, n) for n in names] + [fc.weight, fc.bias]
    pd = [p.detach().to(dev).contiguous() for p in ps]
    gd = [torch.zeros_like(p) for p in pd]
    parr = (C.c_void_p * 10)(*[p.data_ptr() for p in pd])
    garr = (C.c_void_p * 10)(*[p.data_ptr() for p in gd])
    ws = torch.empty(lib.hhrl_lstm_workspace_floats(B, T, o, H), device=dev)
    od = torch.zeros(B, H + 2, device=dev)
    st = L.stream_ptr()
    obsd, doutd = obs.to(dev), dout.to(dev)
    L.check(lib.hhrl_lstm_forward(obsd.data_ptr(), ld, parr, od.data_ptr(), H + 2, ws.data_ptr(), B, T, o, H, st))
    scaled_close("out", od[:, :H].cpu().numpy(), out.detach().numpy(), atol_rel=5e-6)
    L.check(lib.hhrl_lstm_backward(parr, doutd.data_ptr(), H + 2, garr, ws.data_ptr(), B, T, o, H, st))
    for n, p, gdev in zip(names + ["fc.weight", "fc.bias"], ps, gd):
        want = p.grad if p.grad is not None else torch.zeros_like(p)
        scaled_close("grad " + n, gdev.cpu().numpy(), want.numpy(), atol_rel=1e-5)
