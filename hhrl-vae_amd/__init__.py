"""hhrl_vae_amd — MI355X-native learning hot path of HHRL-VAE.

Importable as `hhrl_vae_amd` (the repository directory is `hhrl-vae_amd/`; the
top-level `hhrl_vae_amd.py` shim maps the importable name onto it).

    from hhrl_vae_amd.algorithm import Encoder, WorkerTD3, HATD3Triple, ...
"""
__version__ = "0.1.0"

from . import _lib  # noqa: F401


def lib_path() -> str:
    return _lib.LIB_PATH
