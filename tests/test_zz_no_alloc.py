"""hhrl_vae_train_step_noalloc (the training step whose workspace is a device array built into the library,
for callers that cannot allocate) against hhrl_vae_train_step with a caller-provided workspace, through the
C ABI: same gradients and loss bit for bit, on two calls in a row of which the second is smaller (the array
keeps the first call's contents), then a larger one again; sizes beyond the array are rejected."""
import ctypes

import numpy as np
import pytest
import torch

import gen

pytestmark = pytest.mark.gpu


def _setup(g):
    from hhrl_vae_amd import _lib
    from hhrl_vae_amd.flat import FlatBuffer, Layout
    from hhrl_vae_amd.vae import vae_shapes
    lib = _lib.load()
    dev = torch.device("cuda:0")
    layout = Layout(vae_shapes(g))
    params = FlatBuffer(layout, dev)
    params.load(gen.gen_state(vae_shapes(g), 900 + g))
    offsets = (ctypes.c_long * 24)(*[layout.offsets[n] for n in layout.names])
    return lib, _lib, dev, layout, params, offsets


def _both(lib, L, dev, layout, params, offsets, B, N, g, seed):
    x = torch.from_numpy(gen.gen_lanes(B, N, seed)).to(dev)
    eps = torch.randn(B, g, generator=torch.Generator().manual_seed(seed)).to(dev)
    st = L.stream_ptr()
    out = []
    # reference: the existing entry point with a fresh zero-filled workspace
    n_ws = lib.hhrl_vae_workspace_floats(B, N, g, layout.size)
    ws = torch.zeros(n_ws, dtype=torch.float32, device=dev)
    g_ref = torch.full((layout.size,), 7.0, dtype=torch.float32, device=dev)
    loss_ref = torch.zeros(1, dtype=torch.float32, device=dev)
    L.check(lib.hhrl_vae_train_step(params.base_ptr, g_ref.data_ptr(), offsets, layout.size, x.data_ptr(), eps.data_ptr(),
                                    B, N, g, 1.0 / (B * N), loss_ref.data_ptr(), ws.data_ptr(), st))
    g_new = torch.full((layout.size,), 7.0, dtype=torch.float32, device=dev)
    loss_new = torch.zeros(1, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated(dev)
    L.check(lib.hhrl_vae_train_step_noalloc(params.base_ptr, g_new.data_ptr(), offsets, layout.size, x.data_ptr(),
                                            eps.data_ptr(), B, N, g, 1.0 / (B * N), loss_new.data_ptr(), st))
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated(dev) == before
    return g_ref.cpu().numpy(), loss_ref.cpu().numpy(), g_new.cpu().numpy(), loss_new.cpu().numpy(), n_ws


@pytest.mark.parametrize("g", [4, 16])
def test_noalloc_step_equals_workspace_step(g):
    env = _setup(g)
    lib = env[0]
    sizes = [(64, 23), (37, 5), (64, 59), (1, 1), (64, 7)]     # bigger, smaller, the largest chunk of the script, smallest
    for i, (B, N) in enumerate(sizes):
        g_ref, l_ref, g_new, l_new, n_ws = _both(*env, B, N, g, 50 + i)
        assert n_ws <= 2 * 1024 * 1024
        assert np.isfinite(l_ref).all() and l_ref[0] != 0.0
        assert np.array_equal(l_new, l_ref), (B, N, l_new, l_ref)
        assert np.array_equal(g_new, g_ref), (B, N, float(np.abs(g_new - g_ref).max()))


def test_noalloc_step_rejects_what_the_array_cannot_hold():
    lib, L, dev, layout, params, offsets = _setup(4)
    B, N = 4096, 7
    assert lib.hhrl_vae_workspace_floats(B, N, 4, layout.size) > 2 * 1024 * 1024
    x = torch.zeros(B, N, 2, device=dev)
    eps = torch.zeros(B, 4, device=dev)
    gr = torch.zeros(layout.size, device=dev)
    loss = torch.zeros(1, device=dev)
    rc = lib.hhrl_vae_train_step_noalloc(params.base_ptr, gr.data_ptr(), offsets, layout.size, x.data_ptr(), eps.data_ptr(),
                                         B, N, 4, 1.0 / (B * N), loss.data_ptr(), L.stream_ptr())
    assert rc == 1                                              # HHRL_ERR_ARG
    assert b"scratch" in lib.hhrl_last_error()
    with pytest.raises(L.HipError):
        L.check(rc)
