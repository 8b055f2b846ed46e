"""The reference's `algorithm` module surface (src/algorithm.py), backed by
the HIP kernels: `from hhrl_vae_amd.algorithm import Encoder, WorkerTD3, ...`
is the drop-in for `from algorithm import ...` in agent.py / VAE_train.py."""
from .td3 import HATD3, HATD3Double, HATD3Triple, ManagerTD3, TD3Single, WorkerTD3  # noqa: F401
from .vae import Encoder  # noqa: F401

__all__ = ["Encoder", "TD3Single", "HATD3Double", "HATD3", "HATD3Triple", "ManagerTD3", "WorkerTD3"]
