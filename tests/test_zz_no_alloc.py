"""hhrl_vae_train_step_noalloc (the training step whose workspace is a device array built into the library,
for callers that cannot allocate) against hhrl_vae_train_step with a caller-provided workspace, through the
C ABI: same gradients and loss bit for bit, on two calls in a row of which the second is smaller (the array
keeps the first call's contents), then a larger one again, the same shape twice in a row, and a call captured
into a HIP graph and replayed after an eager call of another shape; sizes beyond the array are rejected.
"Allocates nothing" is checked on the device's free memory (hipMemGetInfo), which sees any hipMalloc in the
process, not on torch's allocator statistics."""
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


def _both(lib, L, dev, layout, params, offsets, B, N, g, seed, check_free=True):
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
    before, before_torch = torch.cuda.mem_get_info(dev)[0], torch.cuda.memory_allocated(dev)
    L.check(lib.hhrl_vae_train_step_noalloc(params.base_ptr, g_new.data_ptr(), offsets, layout.size, x.data_ptr(),
                                            eps.data_ptr(), B, N, g, 1.0 / (B * N), loss_new.data_ptr(), st))
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated(dev) == before_torch
    if check_free:                              # free device memory as the driver reports it: any hipMalloc would show
        assert torch.cuda.mem_get_info(dev)[0] == before, (before, torch.cuda.mem_get_info(dev)[0])
    return g_ref.cpu().numpy(), loss_ref.cpu().numpy(), g_new.cpu().numpy(), loss_new.cpu().numpy(), n_ws


@pytest.mark.parametrize("g", [4, 16])
def test_noalloc_step_equals_workspace_step(g):
    env = _setup(g)
    lib = env[0]
    # bigger, smaller, the largest chunk of the script, smallest, then one shape three times with different data
    # (what a training loop does on every step after the first)
    sizes = [(64, 23), (37, 5), (64, 59), (1, 1), (64, 7), (64, 7), (64, 7)]
    _both(*env, 64, 7, g, 49, check_free=False)                # first use on the device: code objects, the runtime's own buffers
    for i, (B, N) in enumerate(sizes):
        g_ref, l_ref, g_new, l_new, n_ws = _both(*env, B, N, g, 50 + i)
        assert n_ws <= 2 * 1024 * 1024
        assert np.isfinite(l_ref).all() and l_ref[0] != 0.0
        assert np.array_equal(l_new, l_ref), (B, N, l_new, l_ref)
        assert np.array_equal(g_new, g_ref), (B, N, float(np.abs(g_new - g_ref).max()))


def test_noalloc_step_captured_then_replayed_after_another_shape():
    """The entry point under HIP graph capture: shape A is captured (after one eager call, which resolves the
    array's address), an eager call of shape B re-carves and dirties the array, then A is replayed on new data:
    bit-equal to the workspace entry point."""
    lib, L, dev, layout, params, offsets = _setup(4)
    g, (B, N), (B2, N2) = 4, (48, 9), (64, 31)
    _both(lib, L, dev, layout, params, offsets, B, N, g, 70, check_free=False)
    x = torch.zeros(B, N, 2, device=dev)
    eps = torch.zeros(B, g, device=dev)
    grads = torch.zeros(layout.size, device=dev)
    loss = torch.zeros(1, device=dev)
    graph = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(graph):
        L.check(lib.hhrl_vae_train_step_noalloc(params.base_ptr, grads.data_ptr(), offsets, layout.size, x.data_ptr(),
                                                eps.data_ptr(), B, N, g, 1.0 / (B * N), loss.data_ptr(), L.stream_ptr()))
    for seed in (71, 72):
        _both(lib, L, dev, layout, params, offsets, B2, N2, g, 80 + seed)          # another shape, eagerly, in between
        xs = torch.from_numpy(gen.gen_lanes(B, N, seed)).to(dev)
        es = torch.randn(B, g, generator=torch.Generator().manual_seed(seed)).to(dev)
        x.copy_(xs)
        eps.copy_(es)
        graph.replay()
        ws = torch.zeros(lib.hhrl_vae_workspace_floats(B, N, g, layout.size), dtype=torch.float32, device=dev)
        g_ref = torch.zeros(layout.size, device=dev)
        l_ref = torch.zeros(1, device=dev)
        L.check(lib.hhrl_vae_train_step(params.base_ptr, g_ref.data_ptr(), offsets, layout.size, xs.data_ptr(), es.data_ptr(),
                                        B, N, g, 1.0 / (B * N), l_ref.data_ptr(), ws.data_ptr(), L.stream_ptr()))
        torch.cuda.synchronize()
        assert torch.equal(loss, l_ref) and float(l_ref) != 0.0
        assert torch.equal(grads, g_ref), float((grads - g_ref).abs().max())


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
