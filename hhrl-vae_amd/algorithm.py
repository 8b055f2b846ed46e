"""The reference's `algorithm` module surface (src/algorithm.py), backed by
the HIP kernels: `from hhrl_vae_amd.algorithm import Encoder, WorkerTD3, ...`
is the drop-in for `from algorithm import ...` in agent.py / VAE_train.py."""
from .vae import Encoder  # noqa: F401
