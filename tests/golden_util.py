"""Helpers to read the golden fixtures (tests/golden/*.npz)."""
from __future__ import annotations

import os
from typing import Dict

import numpy as np

import gen

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name: str) -> Dict[str, np.ndarray]:
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def sub(d: Dict[str, np.ndarray], prefix: str) -> Dict[str, np.ndarray]:
    """All entries under `prefix/`, with the prefix stripped."""
    pre = prefix.rstrip("/") + "/"
    return {k[len(pre):]: v for k, v in d.items() if k.startswith(pre)}


def assert_close(name: str, got, want, rtol=1e-4, atol=1e-6, max_frac=0.0):
    """|got - want| <= atol + rtol * |want| + max_frac * max|want|, element by element.  max_frac is the share of
    the tensor's largest entry that fp32 summation order alone can move an element by (another CPU's BLAS
    blocking sums the same products in another order)."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, "%s: shape %s vs %s" % (name, got.shape, want.shape)
    err = np.abs(got - want)
    tol = atol + rtol * np.abs(want) + max_frac * (float(np.abs(want).max()) if want.size else 0.0)
    if not np.all(err <= tol):
        i = np.unravel_index(np.argmax(err - tol), err.shape) if err.ndim else ()
        raise AssertionError("%s: max excess at %s: got %r want %r (|err| %.3e, tol %.3e); %d/%d elements out"
                             % (name, i, got[i], want[i], err[i], tol[i], int(np.sum(err > tol)), err.size))


def assert_dict_close(name: str, got: Dict[str, np.ndarray], want: Dict[str, np.ndarray], rtol=1e-4, atol=1e-6,
                      digest: bool = False, max_frac=0.0):
    assert set(want).issubset(set(got)), "%s: missing keys %s" % (name, sorted(set(want) - set(got)))
    for k, w in want.items():
        g = np.asarray(got[k])
        if digest:
            d = gen.digest(g)
            # the two leading sums accumulate rounding over many elements: scale atol by count
            n = max(g.size, 1)
            assert_close(name + "/" + k + "[sums]", d[:2], w[:2], rtol=rtol, atol=atol * n)
            assert_close(name + "/" + k + "[samples]", d[2:], w[2:], rtol=rtol, atol=atol, max_frac=max_frac)
        else:
            assert_close(name + "/" + k, g, w, rtol=rtol, atol=atol, max_frac=max_frac)


def td3_inputs(case: str, fx: Dict[str, np.ndarray], shapes_by_net):
    """Initial weights and replay rows of a TD3 case: stored for 'full'
    fixtures, regenerated from the case's seeds for 'digest' ones."""
    mode = str(fx["mode"])
    n_rows = int(fx["n_rows"])
    width = int(fx["row_width"])
    if mode == "full":
        rows = fx["rows"]
        init = {net: sub(fx, "init/" + net) for net in shapes_by_net}
    else:
        rows = gen.gen_rows(n_rows, width, gen.case_seed(case, "rows"))
        init = {net: gen.gen_state(shp, gen.case_seed(case, "init_" + net)) for net, shp in shapes_by_net.items()}
    return init, rows


def adam_path_close(name, got_p, want_p, grads_by_step, lrs, g_floor=2e-6, rtol=1e-4, max_saturated=None, zero_is_exact=False):
    """Parameters after a sequence of Adam steps, element by element.

    Adam normalises the gradient: a step moves a weight by about lr * g/|g| while the moments are young, so a
    gradient element that two fp32 evaluations know only to within dg (their summation noise, taken as
    g_floor * max|g| of the tensor at that step) makes that step uncertain by about lr * dg/|g|, and by a full
    step (at most ~2.2 lr, the largest bias-corrected ratio) where the sign itself is undetermined.  The bar is
    the rtol of the specification plus a scale floor plus the SUM of those per-step terms: it stays tight for
    every element whose gradient is well above the noise and does not grow with the number of steps for them.

    grads_by_step: list of {tensor name: reference gradient at that step}; lrs: the lr of each step.

    An element whose gradient sits in the noise at most steps ends up with an allowance near 2.2 * sum(lr), the
    furthest Adam can move a weight at all: it is then effectively unchecked.  The share of such elements
    (allowance beyond half of that ceiling) over all tensors is returned, and with max_saturated it is asserted
    to stay below that share, so that the check cannot quietly turn into one that passes anything.

    zero_is_exact: a reference gradient that is exactly 0.0 (a weight behind a ReLU that is off for the whole batch, a
    padded column) is taken as structurally zero -- the other implementation must not move that weight at all -- instead of
    as a gradient of unknown sign."""
    acc = {k: 0.0 for k in want_p}
    for grads, lr in zip(grads_by_step, lrs):
        for k in want_p:
            g = np.abs(np.asarray(grads[k], dtype=np.float64))
            dg = g_floor * max(float(g.max()) if g.size else 0.0, 1e-30)
            step = lr * np.minimum(2.2, 6.0 * dg / (g + 1e-30))
            acc[k] = acc[k] + (np.where(g == 0.0, 0.0, step) if zero_is_exact else step)
    ceiling = 2.2 * float(sum(lrs))
    n_sat = n_all = 0
    for k, w in want_p.items():
        wp, gp = np.asarray(w, dtype=np.float64), np.asarray(got_p[k], dtype=np.float64)
        tol = rtol * np.abs(wp) + 2e-6 * float(np.abs(wp).max() if wp.size else 0.0) + acc[k]
        err = np.abs(gp - wp)
        assert np.all(err <= tol), "%s/%s: %d of %d elements out, worst excess %.3e (bar there %.3e)" % (
            name, k, int((err > tol).sum()), err.size, float((err - tol).max()), float(tol.reshape(-1)[np.argmax(err - tol)]))
        if ceiling > 0:
            n_sat += int(np.sum(np.broadcast_to(acc[k], wp.shape) > 0.5 * ceiling))
        n_all += wp.size
    frac = n_sat / max(n_all, 1)
    print("adam_path_close %s: %d of %d elements (%.2f %%) carry an allowance beyond half of Adam's reach over %d steps"
          % (name, n_sat, n_all, 100.0 * frac, len(lrs)))
    if max_saturated is not None:
        assert frac <= max_saturated, "%s: %.1f %% of the elements are effectively unchecked (allowed: %.1f %%)" % (
            name, 100.0 * frac, 100.0 * max_saturated)
    return frac
