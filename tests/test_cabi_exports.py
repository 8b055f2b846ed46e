"""The C-ABI shared library loads on a CPU-only box and exports every symbol
that include/hhrl_hip.h declares; the ctypes prototype table covers the same
set.  No compute call is made here."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "hhrl_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hhrl_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_entry_points():
    syms = declared_symbols()
    assert len(syms) >= 20
    for must in ("hhrl_linear_forward", "hhrl_linear_backward", "hhrl_lstm_forward", "hhrl_lstm_backward",
                 "hhrl_adam_step", "hhrl_polyak_update", "hhrl_vae_elbo", "hhrl_twin_mse"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    import hhrl_vae_amd
    from hhrl_vae_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):            # built by __graft_entry__.build(); build here if absent
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, "declared in hhrl_hip.h but not exported: %s" % missing
    header = open(os.path.join(ROOT, "include", "hhrl_hip.h")).read()
    declared = int(re.search(r"#define\s+HHRL_ABI_VERSION\s+(\d+)", header).group(1))
    assert lib.hhrl_version() == declared == _lib.ABI_VERSION


def test_ctypes_table_matches_header():
    from hhrl_vae_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    lib = _lib.load()
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(lib, name)
        assert fn.restype is res and list(fn.argtypes) == list(args)


def test_product_refuses_to_run_without_gpu():
    """No CPU fallback: constructing a learner without a GPU raises."""
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from hhrl_vae_amd.algorithm import Encoder
    with pytest.raises(RuntimeError):
        Encoder(2, 4)
