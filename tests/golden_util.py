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


def assert_close(name: str, got, want, rtol=1e-4, atol=1e-6):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, "%s: shape %s vs %s" % (name, got.shape, want.shape)
    err = np.abs(got - want)
    tol = atol + rtol * np.abs(want)
    if not np.all(err <= tol):
        i = np.unravel_index(np.argmax(err - tol), err.shape) if err.ndim else ()
        raise AssertionError("%s: max excess at %s: got %r want %r (|err| %.3e, tol %.3e); %d/%d elements out"
                             % (name, i, got[i], want[i], err[i], tol[i], int(np.sum(err > tol)), err.size))


def assert_dict_close(name: str, got: Dict[str, np.ndarray], want: Dict[str, np.ndarray], rtol=1e-4, atol=1e-6,
                      digest: bool = False):
    assert set(want).issubset(set(got)), "%s: missing keys %s" % (name, sorted(set(want) - set(got)))
    for k, w in want.items():
        g = np.asarray(got[k])
        if digest:
            d = gen.digest(g)
            # the two leading sums accumulate rounding over many elements: scale atol by count
            n = max(g.size, 1)
            assert_close(name + "/" + k + "[sums]", d[:2], w[:2], rtol=rtol, atol=atol * n)
            assert_close(name + "/" + k + "[samples]", d[2:], w[2:], rtol=rtol, atol=atol)
        else:
            assert_close(name + "/" + k, g, w, rtol=rtol, atol=atol)


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
