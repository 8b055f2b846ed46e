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
    # activation and bias gradient inside the backward products: a reduction split over workgroups, odd sizes
    (1500, 70, 33, 3), (64, 301, 37, 2), (33, 7, 65, 1),
    # enough 64 x 64 output tiles for the double-buffered block kernel: aligned and odd leading dimensions (ldx = K + 3),
    # partial tiles on both edges, a reduction that ends inside a chunk
    (1024, 393, 512, 1), (1024, 509, 256, 2), (2048, 128, 400, 1), (1000, 300, 401, 3), (4096, 256, 128, 1),
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


@pytest.mark.parametrize("M,K,N,act", [(64, 393, 512, 1), (70, 33, 9, 2), (5000, 40, 24, 3), (70000, 130, 20, 1)])
def test_linear_backward_partial_outputs(env, M, K, N, act):
    """hhrl_linear_backward with some of dx / dw / db left out gives what the full call gives for the rest
    (small products carry the activation and bias gradients themselves, large ones take a pass of their own,
    and a bias gradient without a weight gradient always does)."""
    lib, L, dev = env
    g = torch.Generator().manual_seed(M + K + N)
    t = lambda *s: torch.randn(*s, generator=g).to(dev)
    x, w, b, dy = t(M, K), t(N, K) / K ** 0.5, t(N), t(M, N)
    y = torch.zeros(M, N, device=dev)
    st = L.stream_ptr()
    L.check(lib.hhrl_linear_forward(x.data_ptr(), K, w.data_ptr(), b.data_ptr(), y.data_ptr(), N, M, K, N, act, 0, st))
    z = lambda *s: torch.zeros(*s, device=dev)

    def run(want_dx, want_dw, want_db):
        dx, dw, db, d = z(M, K), z(N, K), z(N), dy.clone()
        L.check(lib.hhrl_linear_backward(x.data_ptr(), K, w.data_ptr(), y.data_ptr(), N, d.data_ptr(), N,
                                         dx.data_ptr() if want_dx else None, K, 0, dw.data_ptr() if want_dw else None,
                                         db.data_ptr() if want_db else None, M, K, N, act, st))
        return dx, dw, db

    dx0, dw0, db0 = run(True, True, True)
    pre = dy * {1: (y > 0).float(), 2: 1 - y * y, 3: y * (1 - y)}[act]
    scaled_close("dx", dx0.cpu().numpy(), (pre @ w).cpu().numpy(), atol_rel=1e-5)
    scaled_close("dw", dw0.cpu().numpy(), (pre.t() @ x).cpu().numpy(), atol_rel=1e-5)
    scaled_close("db", db0.cpu().numpy(), pre.sum(0).cpu().numpy(), atol_rel=1e-5)
    for flags in [(True, False, False), (False, True, True), (False, True, False), (True, False, True), (False, False, True)]:
        dx, dw, db = run(*flags)
        for name, on, got, want in (("dx", flags[0], dx, dx0), ("dw", flags[1], dw, dw0), ("db", flags[2], db, db0)):
            if on:
                scaled_close("%s %s" % (name, flags), got.cpu().numpy(), want.cpu().numpy(), atol_rel=1e-5)
            else:
                assert not got.any(), "%s written although it was not asked for" % name


@pytest.mark.parametrize("M,K,N,act", [(64, 512, 256, 1), (64, 400, 300, 1), (1024, 512, 256, 1), (37, 19, 7, 2), (5, 3, 1, 0)])
def test_linear_backward_row(env, M, K, N, act):
    """hhrl_linear_backward_row: the input gradient of a layer whose output gradient is one row for every sample,
    against autograd of alpha * sum_i <act(x_i W^T + b), row>; written or accumulated, inside a wider buffer."""
    lib, L, dev = env
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g).requires_grad_(True)
    w, b, row = torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g), torch.randn(N, generator=g)
    alpha = -1.0 / M
    y = _act(F.linear(x, w, b), act)
    (alpha * (y * row).sum()).backward()
    d = lambda t: t.detach().to(dev).contiguous()
    yd, wd, rowd = d(y), d(w), d(row)
    dx = torch.full((M, K + 3), 3.0, device=dev)
    st = L.stream_ptr()
    L.check(lib.hhrl_linear_backward_row(wd.data_ptr(), yd.data_ptr(), N, rowd.data_ptr(), alpha, dx.data_ptr(), K + 3, 0, M, K, N, act, st))
    scaled_close("dx", dx[:, :K].cpu().numpy(), x.grad.numpy(), atol_rel=5e-6)
    assert torch.all(dx[:, K:] == 3.0)
    L.check(lib.hhrl_linear_backward_row(wd.data_ptr(), yd.data_ptr(), N, rowd.data_ptr(), alpha, dx.data_ptr(), K + 3, 1, M, K, N, act, st))
    scaled_close("dx2", dx[:, :K].cpu().numpy(), 2 * x.grad.numpy(), atol_rel=5e-6)


@pytest.mark.parametrize("M,K,N,act,c0,nc", [(64, 393, 512, 1, 384, 4), (1024, 393, 512, 1, 389, 4), (33, 20, 9, 2, 0, 1), (7, 5, 3, 0, 2, 3)])
def test_linear_backward_cols(env, M, K, N, act, c0, nc):
    """hhrl_linear_backward_cols: the named columns of the input gradient equal those of hhrl_linear_backward's, the
    others are not touched."""
    lib, L, dev = env
    g = torch.Generator().manual_seed(M + K + N + c0)
    t = lambda *s: torch.randn(*s, generator=g).to(dev)
    w, y, dy = t(N, K) / K ** 0.5, t(M, N), t(M, N)
    if act == 1:
        y = y.clamp_min(0.0)
    elif act == 2:
        y = torch.tanh(y)
    st = L.stream_ptr()
    full = torch.zeros(M, K, device=dev)
    L.check(lib.hhrl_linear_backward(None, 0, w.data_ptr(), y.data_ptr(), N, dy.clone().data_ptr(), N, full.data_ptr(), K, 0, None, None,
                                     M, K, N, act, st))
    part = torch.full((M, K), 5.0, device=dev)
    L.check(lib.hhrl_linear_backward_cols(w.data_ptr(), K, y.data_ptr(), N, dy.data_ptr(), N, part.data_ptr(), K, 0, c0, nc, M, N, act, st))
    scaled_close("cols", part[:, c0:c0 + nc].cpu().numpy(), full[:, c0:c0 + nc].cpu().numpy(), atol_rel=5e-6)
    mask = torch.ones(K, dtype=torch.bool)
    mask[c0:c0 + nc] = False
    assert torch.all(part[:, mask] == 5.0)
    assert lib.hhrl_linear_backward_cols(w.data_ptr(), K, y.data_ptr(), N, dy.data_ptr(), N, part.data_ptr(), K, 0, K - 1, 2, M, N, act, st) != 0


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


@pytest.mark.parametrize("n,off", [(5, 0), (1216898, 0), (38146, 0), (4099, 1)])
def test_adam_polyak_fused_equals_two_launches(env, n, off):
    """hhrl_adam_polyak_step_dev = hhrl_adam_step_dev followed by hhrl_polyak_update, bit for bit, on aligned
    buffers (float4 body + scalar tail) and on a base that is not 16-byte aligned (scalar form)."""
    lib, L, dev = env
    g = torch.Generator().manual_seed(n)
    mk = lambda: torch.randn(n + off, generator=g)
    p0, gr, m0, v0, t0 = mk(), mk() * 0.1, mk() * 0.01, mk().abs() * 1e-3, mk()
    hyper = torch.zeros(8, device=dev)
    L.check(lib.hhrl_adam_stage(hyper.data_ptr(), 1e-3, 0.9, 0.999, 1e-8, 3, 1.0, L.stream_ptr()))
    outs = []
    for fused in (False, True):
        p, grd, m, v, t = (x.to(dev) for x in (p0, gr, m0, v0, t0))
        ptr = lambda x: x.data_ptr() + 4 * off
        if fused:
            L.check(lib.hhrl_adam_polyak_step_dev(ptr(p), ptr(grd), ptr(m), ptr(v), n, hyper.data_ptr(), ptr(t), 0.005, L.stream_ptr()))
        else:
            L.check(lib.hhrl_adam_step_dev(ptr(p), ptr(grd), ptr(m), ptr(v), n, hyper.data_ptr(), L.stream_ptr()))
            L.check(lib.hhrl_polyak_update(ptr(t), ptr(p), n, 0.005, L.stream_ptr()))
        outs.append([x.cpu() for x in (p, m, v, t)])
    for a, b, name in zip(outs[0], outs[1], "pmvt"):
        assert torch.equal(a, b), name
    assert not torch.equal(outs[0][3][off:], t0[off:]) and not torch.equal(outs[0][0][off:], p0[off:])
    if off:
        assert all(float(x[0]) == float(y[0]) for x, y in zip(outs[1], (p0, m0, v0, t0)))
    # and the step itself against the reference formula (torch.optim.Adam semantics at step 3)
    ref_m = 0.9 * m0[off:].double() + 0.1 * gr[off:].double()
    ref_v = 0.999 * v0[off:].double() + 0.001 * gr[off:].double() ** 2
    ref_p = p0[off:].double() - (1e-3 / (1 - 0.9 ** 3)) * ref_m / (ref_v.sqrt() / (1 - 0.999 ** 3) ** 0.5 + 1e-8)
    scaled_close("p", outs[1][0][off:].numpy(), ref_p.numpy(), rtol=1e-5, atol_rel=1e-6)
    scaled_close("t", outs[1][3][off:].numpy(), (0.005 * ref_p + 0.995 * t0[off:].double()).numpy(), rtol=1e-5, atol_rel=1e-6)


def test_adam_beyond_the_caches(env):
    """Buffers of 64 Mi + 5 parameters take the non-temporal form of the Adam kernel (plus the scalar tail):
    two steps against torch.optim.Adam on the CPU."""
    lib, L, dev = env
    n = (64 << 20) + 5
    g = torch.Generator().manual_seed(7)
    p = torch.randn(n, generator=g)
    ref_p = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref_p], lr=1e-3)
    pd = p.to(dev)
    md, vd = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    for step in (1, 2):
        gr = torch.randn(n, generator=g) * 0.1
        ref_p.grad = gr.clone()
        opt.step()
        grd = gr.to(dev)
        L.check(lib.hhrl_adam_step(pd.data_ptr(), grd.data_ptr(), md.data_ptr(), vd.data_ptr(), n, 1e-3, 0.9, 0.999,
                                   1e-8, step, 1.0, L.stream_ptr()))
    scaled_close("p", pd.cpu().numpy(), ref_p.detach().numpy(), rtol=1e-5, atol_rel=1e-7)
    st = opt.state[ref_p]
    scaled_close("m", md.cpu().numpy(), st["exp_avg"].numpy(), rtol=1e-5)
    scaled_close("v", vd.cpu().numpy(), st["exp_avg_sq"].numpy(), rtol=1e-5)


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
    scaled_close("loss", loss.cpu().numpy(), [float((recon + kl).detach())])
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
    # the same two steps as one launch: y stored, the losses overwritten (no zeroing by the caller)
    y2 = torch.zeros(B, 1, device=dev)
    loss2 = torch.full((2,), 7.0, device=dev)
    e1, e2 = torch.zeros(B, 1, device=dev), torch.zeros(B, 1, device=dev)
    L.check(lib.hhrl_td3_target_mse(tq1d.data_ptr(), tq2d.data_ptr(), rd.data_ptr(), 1, 0.95, q1d.data_ptr(), q2d.data_ptr(),
                                    1.0 / B, y2.data_ptr(), loss2.data_ptr(), e1.data_ptr(), e2.data_ptr(), B, st))
    assert torch.equal(y2, yd) and torch.equal(e1, dq1) and torch.equal(e2, dq2)
    scaled_close("mse fused", loss2.cpu().numpy(), [float(l1), float(l2)])


@pytest.mark.parametrize("B", [1, 256, 8192, 9001])
def test_td3_target_mse_equals_two_launches(env, B):
    """hhrl_td3_target_mse against hhrl_td3_target_q + hhrl_twin_mse: one workgroup up to 8192 rows, several with
    a zeroed loss and atomics beyond; strided reward column as in a replay row."""
    lib, L, dev = env
    g = torch.Generator().manual_seed(B)
    st = L.stream_ptr()
    q1, q2, t1, t2 = (torch.randn(B, 1, generator=g).to(dev) for _ in range(4))
    rows = torch.randn(B, 5, generator=g).to(dev)
    r_ptr = rows.data_ptr() + 4 * 3
    y_a, y_b = torch.zeros(B, 1, device=dev), torch.zeros(B, 1, device=dev)
    la, lb = torch.zeros(2, device=dev), torch.full((2,), -3.0, device=dev)
    d1a, d2a, d1b, d2b = (torch.zeros(B, 1, device=dev) for _ in range(4))
    L.check(lib.hhrl_td3_target_q(t1.data_ptr(), t2.data_ptr(), r_ptr, 5, 0.9, y_a.data_ptr(), B, st))
    L.check(lib.hhrl_twin_mse(q1.data_ptr(), q2.data_ptr(), y_a.data_ptr(), 1.0 / B, la.data_ptr(), d1a.data_ptr(), d2a.data_ptr(), B, st))
    L.check(lib.hhrl_td3_target_mse(t1.data_ptr(), t2.data_ptr(), r_ptr, 5, 0.9, q1.data_ptr(), q2.data_ptr(), 1.0 / B,
                                    y_b.data_ptr(), lb.data_ptr(), d1b.data_ptr(), d2b.data_ptr(), B, st))
    assert torch.equal(y_a, y_b) and torch.equal(d1a, d1b) and torch.equal(d2a, d2b)
    # the form for a caller that zeroed the loss itself
    lc = torch.zeros(2, device=dev)
    y_c, d1c, d2c = (torch.zeros(B, 1, device=dev) for _ in range(3))
    L.check(lib.hhrl_td3_target_mse_zeroed(t1.data_ptr(), t2.data_ptr(), r_ptr, 5, 0.9, q1.data_ptr(), q2.data_ptr(), 1.0 / B,
                                           y_c.data_ptr(), lc.data_ptr(), d1c.data_ptr(), d2c.data_ptr(), B, st))
    assert torch.equal(y_c, y_b) and torch.equal(d1c, d1b) and torch.equal(d2c, d2b)
    scaled_close("loss (zeroed by the caller)", lc.cpu().numpy(), la.cpu().numpy(), rtol=1e-5)
    want = rows[:, 3:4] + 0.9 * torch.minimum(t1, t2)
    scaled_close("y", y_b.cpu().numpy(), want.cpu().numpy(), rtol=1e-6)
    scaled_close("loss", lb.cpu().numpy(), la.cpu().numpy(), rtol=1e-5)


@pytest.mark.parametrize("M", [4096, 1000, 64, 5])
def test_mlp3_backward_td3_target_equals_two_calls(env, M):
    """The backward launch of two critic heads that forms the TD3 target, both losses and dL/dQ itself (mse_* fields)
    against hhrl_td3_target_mse followed by the plain backward launch: the same target, the same bits in every
    parameter gradient (the slab rows are summed in a fixed order whatever the batch), the same losses."""
    lib, L, dev = env
    K0, n = 11, 2
    g = torch.Generator().manual_seed(77 + M)
    r = lambda *s: torch.randn(*s, generator=g)
    x = r(M, K0).to(dev)
    rows = r(M, 6).to(dev)                              # the reward is column 2 of a wider row
    q1t, q2t = r(M, 1).to(dev), r(M, 1).to(dev)
    ws = [[(r(128, K0) / K0 ** 0.5).to(dev), r(128).to(dev), (r(128, 128) / 128 ** 0.5).to(dev), r(128).to(dev),
           (r(1, 128) / 128 ** 0.5).to(dev), r(1).to(dev)] for _ in range(n)]
    st = L.stream_ptr()
    out = []
    for own in (False, True):
        descs = (L.Mlp3Desc * n)()
        keep, grads = [], []
        loss, y = torch.zeros(2, device=dev), torch.full((M,), 9.0, device=dev)
        dq = [torch.zeros(M, 1, device=dev) for _ in range(n)]
        for i in range(n):
            d = descs[i]
            d.x, d.ldx, d.K0 = x.data_ptr(), K0, K0
            d.w1, d.b1, d.w2, d.b2, d.w3, d.b3 = [w.data_ptr() for w in ws[i]]
            h1, h2, q = torch.empty(M, 128, device=dev), torch.empty(M, 128, device=dev), torch.empty(M, 1, device=dev)
            d.N3, d.act3, d.h1, d.h2, d.y, d.ldy = 1, 0, h1.data_ptr(), h2.data_ptr(), q.data_ptr(), 1
            gd = [torch.zeros_like(w) for w in ws[i]]
            d.gw1, d.gb1, d.gw2, d.gb2, d.gw3, d.gb3 = [t.data_ptr() for t in gd]
            n_slab = lib.hhrl_mlp3_slab_floats(K0, 1, M)
            slab = torch.full((max(n_slab, 1),), float("nan"), device=dev)
            if n_slab:
                d.gslab, d.gslab_floats = slab.data_ptr(), n_slab
            keep.append((h1, h2, q, slab))
            grads.append(gd)
        L.check(lib.hhrl_mlp3_forward_group(descs, n, M, st))
        if own:
            for i in range(n):
                d = descs[i]
                d.mse_q1t, d.mse_q2t, d.mse_r, d.mse_ldr = q1t.data_ptr(), q2t.data_ptr(), rows.data_ptr() + 4 * 2, 6
                d.mse_gamma, d.mse_inv_n, d.mse_loss, d.mse_target = 0.9, 1.0 / M, loss.data_ptr() + 4 * i, y.data_ptr()
        else:
            L.check(lib.hhrl_td3_target_mse(q1t.data_ptr(), q2t.data_ptr(), rows.data_ptr() + 4 * 2, 6, 0.9, keep[0][2].data_ptr(),
                                            keep[1][2].data_ptr(), 1.0 / M, y.data_ptr(), loss.data_ptr(), dq[0].data_ptr(),
                                            dq[1].data_ptr(), M, st))
            for i in range(n):
                descs[i].dy, descs[i].lddy = dq[i].data_ptr(), 1
        L.check(lib.hhrl_mlp3_backward_group(descs, n, M, st))
        torch.cuda.synchronize()
        out.append((loss, y, grads))
    (la, ya, ga), (lb, yb, gb) = out
    assert torch.equal(ya, yb)
    scaled_close("losses", lb.cpu().numpy(), la.cpu().numpy(), rtol=1e-5)
    for i in range(n):
        for k, (a, b) in enumerate(zip(ga[i], gb[i])):
            assert torch.equal(a, b), "gradient %d of head %d" % (k, i)
    # the form is refused for a wider output or together with a dy
    descs[0].dy, descs[0].lddy = dq[0].data_ptr(), 1
    assert lib.hhrl_mlp3_backward_group(descs, n, M, st) != 0 and b"TD3 target" in lib.hhrl_last_error()


@pytest.mark.parametrize("M,polyak", [(4096, True), (4096, False), (300, True), (64, True)])
def test_mlp3_backward_group_adam_equals_two_calls(env, M, polyak):
    """hhrl_mlp3_backward_group_adam against hhrl_mlp3_backward_group followed by hhrl_adam_step_dev /
    hhrl_adam_polyak_step_dev on flat buffers that hold two networks (the two heads of a critic): the same bits in
    the gradients, parameters, both moments and the target copy (the step rides on the slab sum; without slabs it
    would follow as a launch of its own); padding between the tensors is left alone."""
    lib, L, dev = env
    K0, N3, n = 11, 1, 2
    g = torch.Generator().manual_seed(1000 + M)
    sizes = [128 * K0, 128, 128 * 128, 128, N3 * 128, N3]
    offs, total = [], 0
    for i in range(n):
        for sz in sizes:
            offs.append(total)
            total += -(-sz // 64) * 64
    flat = lambda scale: (torch.randn(total, generator=g) * scale).to(dev)
    p0, m0, v0, t0 = flat(0.1), flat(0.01), flat(0.01).abs(), flat(0.1)
    x = torch.randn(M, K0, generator=g).to(dev)
    dy = [torch.randn(M, N3, generator=g).to(dev) for _ in range(n)]
    hyper = torch.zeros(8, device=dev)
    st = L.stream_ptr()
    L.check(lib.hhrl_adam_stage(hyper.data_ptr(), 1e-3, 0.9, 0.999, 1e-8, 7, 1.0, st))
    results = []
    for fused in (False, True):
        p, m, v, t = p0.clone(), m0.clone(), v0.clone(), t0.clone()
        gr = torch.zeros(total, device=dev)
        descs = (L.Mlp3Desc * n)()
        keep = []
        for i in range(n):
            d = descs[i]
            o = offs[6 * i: 6 * i + 6]
            P_ = lambda buf, k: buf.data_ptr() + 4 * o[k]
            d.x, d.ldx, d.K0 = x.data_ptr(), K0, K0
            d.w1, d.b1, d.w2, d.b2, d.w3, d.b3 = [P_(p, k) for k in range(6)]
            h1, h2, y = torch.empty(M, 128, device=dev), torch.empty(M, 128, device=dev), torch.empty(M, N3, device=dev)
            d.N3, d.act3, d.h1, d.h2, d.y, d.ldy = N3, 0, h1.data_ptr(), h2.data_ptr(), y.data_ptr(), N3
            d.dy, d.lddy, d.dy_const = dy[i].data_ptr(), N3, 0.0
            d.gw1, d.gb1, d.gw2, d.gb2, d.gw3, d.gb3 = [P_(gr, k) for k in range(6)]
            n_slab = lib.hhrl_mlp3_slab_floats(K0, N3, M)
            slab = torch.full((max(n_slab, 1),), float("nan"), device=dev)
            if n_slab:
                d.gslab, d.gslab_floats = slab.data_ptr(), n_slab
            keep.append((h1, h2, y, slab))
        L.check(lib.hhrl_mlp3_forward_group(descs, n, M, st))
        tp = t.data_ptr() if polyak else None
        if fused:
            ride = L.AdamRide(gr.data_ptr(), p.data_ptr(), m.data_ptr(), v.data_ptr(), total, hyper.data_ptr(), tp, 0.005)
            L.check(lib.hhrl_mlp3_backward_group_adam(descs, n, M, C.byref(ride), st))
        else:
            L.check(lib.hhrl_mlp3_backward_group(descs, n, M, st))
            if polyak:
                L.check(lib.hhrl_adam_polyak_step_dev(p.data_ptr(), gr.data_ptr(), m.data_ptr(), v.data_ptr(), total,
                                                      hyper.data_ptr(), tp, 0.005, st))
            else:
                L.check(lib.hhrl_adam_step_dev(p.data_ptr(), gr.data_ptr(), m.data_ptr(), v.data_ptr(), total, hyper.data_ptr(), st))
        torch.cuda.synchronize()
        results.append((gr, p, m, v, t))
    names = ("gradients", "parameters", "exp_avg", "exp_avg_sq", "target")
    covered = torch.zeros(total, dtype=torch.bool, device=dev)
    for o, sz in zip(offs, sizes * n):
        covered[o: o + sz] = True
    for name, a, b in zip(names, results[0], results[1]):
        # slab sums are the same bits on every run; outside the tensors the riding step touches nothing, the
        # stand-alone one moves padding by its (nonzero, random here) moments
        assert torch.equal(a[covered], b[covered]), name
    assert not torch.equal(results[1][1], p0) and (polyak != torch.equal(results[1][4], t0))
    assert torch.equal(results[1][1][~covered], p0[~covered])
    # a gradient tensor outside the flat buffer is refused
    ride = L.AdamRide(gr.data_ptr() + 4 * 64, p.data_ptr(), m.data_ptr(), v.data_ptr(), total - 64, hyper.data_ptr(), None, 0.0)
    assert lib.hhrl_mlp3_backward_group_adam(descs, n, M, C.byref(ride), st) != 0
    assert b"outside" in lib.hhrl_last_error()


@pytest.mark.parametrize("B,T,o,H", [(9, 5, 7, 12), (64, 10, 13, 128), (64, 10, 25, 128), (3, 1, 4, 8), (130, 4, 5, 8),
                                     # H = 128 with o <= 32 runs the whole window in one launch (four batch rows per workgroup):
                                     # a partial last group of rows, one step, more row groups than workgroups, the widest
                                     # observation it takes, and the first width that goes back to the per-step kernels
                                     (7, 10, 13, 128), (1, 1, 3, 128), (1101, 3, 32, 128), (64, 10, 33, 128)])
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


# ------------------------------------------------------------------ grouped / fused entry points
@pytest.mark.parametrize("B,T,H,obs_dims", [(64, 10, 128, (13, 13, 25)), (33, 4, 32, (5, 29)), (7, 3, 8, (4, 6, 3)),
                                            (130, 6, 128, (13, 32, 5)), (350, 2, 128, (1, 16, 17, 9, 9, 9, 9, 9, 9)),
                                            # several passes per workgroup (an odd number of row groups, a partial last group;
                                            # both observation widths; one module with more groups than workgroups)
                                            (1098, 5, 128, (13, 25, 32)), (2001, 3, 128, (13,))])
def test_lstm_group_matches_torch(env, B, T, H, obs_dims):
    """hhrl_lstm_forward_group / _backward_group: several LSTM modules with different observation
    widths in one call, each against torch.nn.LSTM + Linear."""
    lib, L, dev = env
    torch.manual_seed(B + T + H + sum(obs_dims))
    n = len(obs_dims)
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0", "weight_ih_l0_reverse", "weight_hh_l0_reverse",
             "bias_ih_l0_reverse", "bias_hh_l0_reverse"]
    descs = (L.LstmDesc * n)()
    keep, refs = [], []
    lddo = H + 5
    for i, o in enumerate(obs_dims):
        ref = torch.nn.LSTM(o, H, 1, batch_first=True, bidirectional=True)
        fc = torch.nn.Linear(2 * H, H)
        ld = T * o + 1 + i
        obs = torch.randn(B, ld)
        out = fc(ref(obs[:, :T * o].reshape(B, T, o))[0][:, -1, :])
        dout = torch.randn(B, lddo)
        out.backward(dout[:, :H])
        ps = [getattr(ref, k) for k in names] + [fc.weight, fc.bias]
        pd = [p.detach().to(dev).contiguous() for p in ps]
        gd = [torch.zeros_like(p) for p in pd]
        parr = (C.c_void_p * 10)(*[p.data_ptr() for p in pd])
        garr = (C.c_void_p * 10)(*[p.data_ptr() for p in gd])
        ws = torch.empty(lib.hhrl_lstm_workspace_floats(B, T, o, H), device=dev)
        od = torch.zeros(B, H + 2, device=dev)
        obsd, doutd = obs.to(dev), dout.to(dev)
        d = descs[i]
        d.obs, d.ldobs, d.o = obsd.data_ptr(), ld, o
        d.params, d.grads = C.cast(parr, C.c_void_p), C.cast(garr, C.c_void_p)
        d.out, d.ldo, d.d_out, d.lddo, d.ws = od.data_ptr(), H + 2, doutd.data_ptr(), lddo, ws.data_ptr()
        keep.append((pd, gd, parr, garr, ws, od, obsd, doutd))
        refs.append((out.detach(), ps))
    st = L.stream_ptr()
    L.check(lib.hhrl_lstm_forward_group(descs, n, B, T, H, st))
    L.check(lib.hhrl_lstm_backward_group(descs, n, B, T, H, st))
    for i in range(n):
        out, ps = refs[i]
        scaled_close("out[%d]" % i, keep[i][5][:, :H].cpu().numpy(), out.numpy(), atol_rel=5e-6)
        for k, p, gdev in zip(names + ["fc.weight", "fc.bias"], ps, keep[i][1]):
            want = p.grad if p.grad is not None else torch.zeros_like(p)
            scaled_close("grad[%d] %s" % (i, k), gdev.cpu().numpy(), want.numpy(), atol_rel=1e-5)


@pytest.mark.parametrize("B", [64, 1098])
def test_lstm_forward_without_saved_steps(env, B):
    """hhrl_lstm_desc.no_backward: modules flagged that way give bit for bit the output of an unflagged evaluation
    and leave their workspace alone apart from the fc input (the per-step gates, cell and hidden states and the
    time-major observations are not written); unflagged modules of the same launch are unaffected, whatever share
    of the launch's workgroups each kind gets."""
    lib, L, dev = env
    T, H, obs_dims, flags = 10, 128, (13, 25, 13, 32, 7), (1, 0, 1, 1, 0)
    n = len(obs_dims)
    g = torch.Generator().manual_seed(B)
    r = lambda *s: (torch.randn(*s, generator=g) * 0.3).to(dev)
    st = L.stream_ptr()
    mods = []
    for o in obs_dims:
        ps = [r(4 * H, o), r(4 * H, H), r(4 * H), r(4 * H), r(4 * H, o), r(4 * H, H), r(4 * H), r(4 * H), r(H, 2 * H), r(H)]
        mods.append((o, ps, (C.c_void_p * 10)(*[p.data_ptr() for p in ps]), r(B, T * o + 2)))

    def run(flagged):
        descs, keep = (L.LstmDesc * n)(), []
        for i, (o, ps, parr, obs) in enumerate(mods):
            ws = torch.full((lib.hhrl_lstm_workspace_floats(B, T, o, H),), 7.0, device=dev)
            out = torch.zeros(B, H, device=dev)
            d = descs[i]
            d.obs, d.ldobs, d.o, d.params = obs.data_ptr(), T * o + 2, o, C.cast(parr, C.c_void_p)
            d.out, d.ldo, d.ws, d.no_backward = out.data_ptr(), H, ws.data_ptr(), (flags[i] if flagged else 0)
            keep.append((ws, out))
        L.check(lib.hhrl_lstm_forward_group(descs, n, B, T, H, st))
        torch.cuda.synchronize()
        return keep

    full, part = run(False), run(True)
    for i in range(n):
        assert torch.equal(full[i][1], part[i][1]), "module %d: output differs" % i
        if flags[i]:
            untouched = int((part[i][0] == 7.0).sum())
            assert untouched >= part[i][0].numel() - 2 * B * H, "module %d: %d of %d workspace floats left alone" % (
                i, untouched, part[i][0].numel())
            assert untouched < part[i][0].numel(), "the fc input must have been written"
        else:
            assert torch.equal(full[i][0], part[i][0]), "module %d: saved steps differ" % i


def test_lstm_group_argument_errors(env):
    lib, L, dev = env
    descs = (L.LstmDesc * 1)()
    assert lib.hhrl_lstm_forward_group(descs, 0, 4, 2, 8, None) != 0
    assert lib.hhrl_lstm_forward_group(descs, 17, 4, 2, 8, None) != 0
    assert lib.hhrl_lstm_forward_group(descs, 1, 4, 2, 8, None) != 0          # null pointers inside the descriptor
    assert b"null" in lib.hhrl_last_error()


@pytest.mark.parametrize("M", [64, 37, 1024])
def test_linear_groups_match_single_calls(env, M):
    """hhrl_linear_forward_group / _backward_group against one hhrl_linear_* call per layer."""
    lib, L, dev = env
    g = torch.Generator().manual_seed(M)
    shapes = [(18, 128, 1), (18, 128, 1), (393, 512, 1), (128, 1, 0), (300, 4, 2)]      # (K, N, act)
    n = len(shapes)
    fd, bd = (L.LinearFwdDesc * n)(), (L.LinearBwdDesc * n)()
    keep = []
    st = L.stream_ptr()
    dxs = torch.zeros(M, 18, device=dev)                       # layers 0 and 1 share an input gradient (second accumulates)
    for i, (K, N, act) in enumerate(shapes):
        t = lambda *s: torch.randn(*s, generator=g).to(dev)
        x, w, b, dy = t(M, K + 2), t(N, K) / K ** 0.5, t(N), t(M, N)
        y1, y2 = torch.zeros(M, N + 1, device=dev), torch.zeros(M, N + 1, device=dev)
        L.check(lib.hhrl_linear_forward(x.data_ptr(), K + 2, w.data_ptr(), b.data_ptr(), y1.data_ptr(), N + 1, M, K, N, act, 0, st))
        dy1, dy2 = dy.clone(), dy.clone()
        dx1 = torch.zeros(M, K, device=dev)
        dx2 = dxs if i < 2 else torch.zeros(M, K, device=dev)
        dw1, dw2, db1, db2 = (torch.zeros(N, K, device=dev), torch.zeros(N, K, device=dev), torch.zeros(N, device=dev),
                              torch.zeros(N, device=dev))
        L.check(lib.hhrl_linear_backward(x.data_ptr(), K + 2, w.data_ptr(), y1.data_ptr(), N + 1, dy1.data_ptr(), N, dx1.data_ptr(), K, 0,
                                         dw1.data_ptr(), db1.data_ptr(), M, K, N, act, st))
        f, bw = fd[i], bd[i]
        f.x, f.ldx, f.w, f.b, f.y, f.ldy, f.K, f.N, f.act, f.accumulate = x.data_ptr(), K + 2, w.data_ptr(), b.data_ptr(), y2.data_ptr(), N + 1, K, N, act, 0
        bw.x, bw.ldx, bw.w, bw.y, bw.ldy, bw.dy, bw.lddy = x.data_ptr(), K + 2, w.data_ptr(), y2.data_ptr(), N + 1, dy2.data_ptr(), N
        bw.dx, bw.lddx, bw.dx_accumulate, bw.dw, bw.db, bw.K, bw.N, bw.act = dx2.data_ptr(), K, 1 if i == 1 else 0, dw2.data_ptr(), db2.data_ptr(), K, N, act
        keep.append((x, w, b, y1, y2, dy1, dy2, dx1, dx2, dw1, dw2, db1, db2))
    L.check(lib.hhrl_linear_forward_group(fd, n, M, st))
    L.check(lib.hhrl_linear_backward_group(bd, n, M, st))
    for i, (x, w, b, y1, y2, dy1, dy2, dx1, dx2, dw1, dw2, db1, db2) in enumerate(keep):
        scaled_close("y[%d]" % i, y2.cpu().numpy(), y1.cpu().numpy())
        scaled_close("dw[%d]" % i, dw2.cpu().numpy(), dw1.cpu().numpy(), atol_rel=1e-5)
        scaled_close("db[%d]" % i, db2.cpu().numpy(), db1.cpu().numpy(), atol_rel=1e-5)
        if i >= 2:
            scaled_close("dx[%d]" % i, dx2.cpu().numpy(), dx1.cpu().numpy(), atol_rel=1e-5)
    scaled_close("dx[0]+dx[1]", dxs.cpu().numpy(), (keep[0][7] + keep[1][7]).cpu().numpy(), atol_rel=1e-5)


def test_copy2d_group(env):
    lib, L, dev = env
    M = 70
    src = torch.randn(M, 40, device=dev)
    dst = torch.zeros(M, 30, device=dev)
    blocks = [(0, 5, 3), (10, 0, 2), (20, 12, 11), (39, 29, 1)]            # (source column, destination column, width)
    arr = (L.CopyDesc * len(blocks))()
    for d, (sc, dc, w) in zip(arr, blocks):
        d.inp, d.ldi, d.out, d.ldo, d.C = src.data_ptr() + 4 * sc, 40, dst.data_ptr() + 4 * dc, 30, w
    L.check(lib.hhrl_copy2d_group(arr, len(blocks), M, L.stream_ptr()))
    want = torch.zeros(M, 30, device=dev)
    for sc, dc, w in blocks:
        want[:, dc:dc + w] = src[:, sc:sc + w]
    assert torch.equal(dst, want)
    assert lib.hhrl_copy2d_group(arr, 0, M, None) != 0


def test_gather_copy_group(env):
    """hhrl_gather_copy_group: gathering blocks read row idx[r] of the table (row numbers outside it clamped), plain
    blocks row r, zero fills ride along; everything outside the written ranges stays as it was."""
    lib, L, dev = env
    M, W, n_in = 37, 29, 100
    g = torch.Generator().manual_seed(5)
    table = torch.randn(n_in, W, generator=g).to(dev)
    plain = torch.randn(M, 11, generator=g).to(dev)
    idx = torch.randint(0, n_in, (M,), generator=g, dtype=torch.int32)
    idx[3], idx[7] = -5, n_in + 9
    idx_d = idx.to(dev)
    rows = torch.full((M, W), 9.0, device=dev)
    a = torch.full((M, 20), 9.0, device=dev)
    b = torch.full((M, 8), 9.0, device=dev)
    z = torch.full((70,), 9.0, device=dev)
    d = (L.GCopyDesc * 5)()
    specs = [(table.data_ptr(), W, rows.data_ptr(), W, W, 1),                      # the rows themselves
             (table.data_ptr() + 4 * 5, W, a.data_ptr() + 4 * 2, 20, 13, 1),        # columns 5..17 of the gathered rows
             (table.data_ptr() + 4 * 28, W, b.data_ptr(), 8, 1, 1),                 # the last column
             (plain.data_ptr() + 4 * 1, 11, b.data_ptr() + 4 * 3, 8, 4, 0),         # a plain copy
             (None, 0, z.data_ptr() + 4 * 3, 0, 61, 0)]                             # a zero fill
    for k, (inp, ldi, out, ldo, Cc, ga) in enumerate(specs):
        d[k].inp, d[k].ldi, d[k].out, d[k].ldo, d[k].C, d[k].gather = inp, ldi, out, ldo, Cc, ga
    L.check(lib.hhrl_gather_copy_group(d, 5, M, idx_d.data_ptr(), n_in, L.stream_ptr()))
    want = table[idx_d.long().clamp(0, n_in - 1)]
    assert torch.equal(rows, want)
    assert torch.equal(a[:, 2:15], want[:, 5:18]) and torch.all(a[:, :2] == 9.0) and torch.all(a[:, 15:] == 9.0)
    assert torch.equal(b[:, 0], want[:, 28]) and torch.equal(b[:, 3:7], plain[:, 1:5])
    assert torch.all(b[:, 1:3] == 9.0) and torch.all(b[:, 7] == 9.0)
    assert torch.all(z[3:64] == 0.0) and torch.all(z[:3] == 9.0) and torch.all(z[64:] == 9.0)
    assert lib.hhrl_gather_copy_group(d, 5, M, None, n_in, L.stream_ptr()) != 0 and b"row numbers" in lib.hhrl_last_error()


@pytest.mark.parametrize("M,K0,N3,act3", [(64, 17, 1, 2), (64, 18, 1, 0), (37, 64, 32, 3), (4096, 11, 4, 2), (1, 1, 1, 0),
                                            (1000, 33, 17, 3), (20000, 18, 1, 0)])
def test_mlp3_fused_matches_torch(env, M, K0, N3, act3):
    """hhrl_mlp3_forward_group / _backward_group (two networks side by side) against torch autograd:
    parameter gradients, input gradient, and the constant-upstream-gradient form."""
    lib, L, dev = env
    assert lib.hhrl_mlp3_supported(K0, 128, 128, N3) == 1
    assert lib.hhrl_mlp3_supported(K0, 400, 300, N3) == 0 and lib.hhrl_mlp3_supported(65, 128, 128, N3) == 0
    g = torch.Generator().manual_seed(M + K0 + N3)
    n = 2
    descs = (L.Mlp3Desc * n)()
    keep, refs = [], []
    ldx, ldy = K0 + 3, N3 + 2
    for i in range(n):
        r = lambda *s: torch.randn(*s, generator=g)
        x = r(M, ldx)
        ws = [r(128, K0) / K0 ** 0.5, r(128), r(128, 128) / 128 ** 0.5, r(128), r(N3, 128) / 128 ** 0.5, r(N3)]
        dy = r(M, N3)
        xt = x[:, :K0].clone().requires_grad_(True)
        wt = [w.clone().requires_grad_(True) for w in ws]
        y = _act(F.linear(F.relu(F.linear(F.relu(F.linear(xt, wt[0], wt[1])), wt[2], wt[3])), wt[4], wt[5]), act3)
        const = i == 1                                   # second network: dL/dy is a constant (gradient of a mean)
        (y.sum() * -0.25 if const else (y * dy).sum()).backward()
        xd, dyd = x.to(dev), dy.to(dev).contiguous()
        wd = [w.to(dev).contiguous() for w in ws]
        gd = [torch.zeros_like(w) for w in wd]
        h1, h2 = torch.empty(M, 128, device=dev), torch.empty(M, 128, device=dev)
        yd = torch.full((M, ldy), 7.0, device=dev)
        dxd = torch.zeros(M, K0 + 1, device=dev)
        d = descs[i]
        d.x, d.ldx, d.K0 = xd.data_ptr(), ldx, K0
        d.w1, d.b1, d.w2, d.b2, d.w3, d.b3 = [w.data_ptr() for w in wd]
        d.N3, d.act3, d.h1, d.h2, d.y, d.ldy = N3, act3, h1.data_ptr(), h2.data_ptr(), yd.data_ptr(), ldy
        d.dy, d.lddy, d.dy_const = (None, 0, -0.25) if const else (dyd.data_ptr(), N3, 0.0)
        d.gw1, d.gb1, d.gw2, d.gb2, d.gw3, d.gb3 = [t.data_ptr() for t in gd]
        d.dx, d.lddx = dxd.data_ptr(), K0 + 1
        n_slab = lib.hhrl_mlp3_slab_floats(K0, N3, M)
        assert n_slab > 0
        slab = torch.full((max(n_slab, 1),), float("nan"), device=dev)    # every slab entry read is one the kernel wrote
        if n_slab:
            d.gslab, d.gslab_floats = slab.data_ptr(), n_slab
        keep.append((xd, dyd, wd, gd, h1, h2, yd, dxd, slab))
        refs.append((y.detach(), xt.grad, [w.grad for w in wt]))
    st = L.stream_ptr()
    # network 0 also hands its output through the target-policy smoothing, into a wider row
    sm_eps = (torch.randn(M, N3, generator=g) * 3).to(dev)
    sm_out = torch.full((M, N3 + 3), 5.0, device=dev)
    descs[0].smooth_eps, descs[0].y2, descs[0].ldy2 = sm_eps.data_ptr(), sm_out.data_ptr() + 4 * 2, N3 + 3
    L.check(lib.hhrl_mlp3_forward_group(descs, n, M, st))
    L.check(lib.hhrl_mlp3_backward_group(descs, n, M, st))
    sm_want = torch.full((M, N3 + 3), 5.0, device=dev)
    L.check(lib.hhrl_td3_smooth_action(keep[0][6].data_ptr(), ldy, sm_eps.data_ptr(), sm_want.data_ptr() + 4 * 2, N3 + 3, M, N3, st))
    assert torch.equal(sm_out, sm_want)
    for i in range(n):
        y, dx, gws = refs[i]
        xd, dyd, wd, gd, h1, h2, yd, dxd, _ = keep[i]
        scaled_close("y[%d]" % i, yd[:, :N3].cpu().numpy(), y.numpy(), atol_rel=5e-6)
        assert torch.all(yd[:, N3:] == 7.0), "forward wrote outside its column slice"
        scaled_close("dx[%d]" % i, dxd[:, :K0].cpu().numpy(), dx.numpy(), atol_rel=1e-5)
        assert torch.all(dxd[:, K0:] == 0.0)
        for k, (got, want) in enumerate(zip(gd, gws)):
            scaled_close("grad[%d][%d]" % (i, k), got.cpu().numpy(), want.numpy(), atol_rel=1e-5)
    if n_slab:
        # parameter gradients handed over through slabs are summed in a fixed order: the same bits on every run,
        # and close to what the atomics path (no slab given) adds up
        first = [[t.clone() for t in keep[i][3]] for i in range(n)]
        for i in range(n):
            for t in keep[i][3]:
                t.zero_()
        L.check(lib.hhrl_mlp3_backward_group(descs, n, M, st))
        for i in range(n):
            assert all(torch.equal(a, b) for a, b in zip(first[i], keep[i][3])), "slab path is not reproducible"
            for t in keep[i][3]:
                t.zero_()
            descs[i].gslab, descs[i].gslab_floats = None, 0
        L.check(lib.hhrl_mlp3_backward_group(descs, n, M, st))
        for i in range(n):
            for k, (got, want) in enumerate(zip(keep[i][3], refs[i][2])):
                scaled_close("atomics grad[%d][%d]" % (i, k), got.cpu().numpy(), want.numpy(), atol_rel=1e-5)
    # forward and input gradient for a constant dL/dy in ONE launch (network 1): same y and dx as the two calls above
    d = descs[1]
    y_two, dx_two = keep[1][6].clone(), keep[1][7].clone()
    keep[1][6].fill_(7.0)
    keep[1][7].zero_()
    d.forward_dx = 1
    L.check(lib.hhrl_mlp3_forward_group(C.pointer(d), 1, M, st))
    d.forward_dx = 0
    assert torch.equal(keep[1][6], y_two)
    scaled_close("dx, one launch", keep[1][7].cpu().numpy(), dx_two.cpu().numpy(), atol_rel=1e-6)
    # input gradient only (no parameter gradients): same dx, gradient buffers untouched
    before = [t.clone() for t in keep[0][3]]
    d = descs[0]
    d.gw1 = d.gb1 = d.gw2 = d.gb2 = d.gw3 = d.gb3 = None
    keep[0][7].zero_()
    L.check(lib.hhrl_mlp3_backward_group(descs, 1, M, st))
    scaled_close("dx only", keep[0][7][:, :K0].cpu().numpy(), refs[0][1].numpy(), atol_rel=1e-5)
    assert all(torch.equal(a, b) for a, b in zip(before, keep[0][3]))


def test_randn_fill_is_a_reproducible_normal_stream(env):
    """hhrl_randn_fill: the same (seed, offset) gives the same values, consecutive fills continue one stream whatever
    their lengths and alignment, other seeds give other values, and the values are standard normal (moments and a
    Kolmogorov-Smirnov distance over 4 Mi draws; tails beyond 4.5 sigma exist and are finite)."""
    from scipy import stats
    lib, L, dev = env
    st = L.stream_ptr()
    n = 1 << 22

    def fill(n, seed, off):
        out = torch.full((n + 8,), 77.0, device=dev)
        L.check(lib.hhrl_randn_fill(out.data_ptr(), n, seed, off, st))
        assert torch.all(out[n:] == 77.0), "wrote past the end"
        return out[:n].cpu().numpy()

    a = fill(n, 3407, 0)
    assert np.array_equal(a, fill(n, 3407, 0))
    for off, m in ((0, 5), (5, 1000), (1005, 3), (1008, 4099), (5107, 1)):          # pieces at odd offsets
        assert np.array_equal(fill(m, 3407, off), a[off:off + m])
    b = fill(n, 3408, 0)
    assert not np.array_equal(a, b) and abs(np.corrcoef(a, b)[0, 1]) < 3e-3
    assert np.all(np.isfinite(a))
    x = a.astype(np.float64)
    assert abs(x.mean()) < 3e-3 and abs(x.var() - 1.0) < 5e-3
    assert abs(stats.skew(x)) < 1e-2 and abs(stats.kurtosis(x)) < 2e-2
    assert stats.kstest(x[:1 << 20], "norm").statistic < 2.5e-3
    assert abs(np.corrcoef(x[:-1], x[1:])[0, 1]) < 3e-3                              # neighbours (a Box-Muller pair) uncorrelated
    assert 20 < np.sum(np.abs(x) > 4.5) < 90                                         # expected 28.5 of 4 Mi
