"""One rank of the world-size-2 data-parallel tests (tests/test_zy_dp_world2.py starts two of these as fresh
child processes).  Both ranks use cuda:0 and exchange gradients over the gloo backend: the PRODUCT classes run
with a real group of two, each on its shard of a global minibatch, and write what they end up with.

    python tests/dp_worker.py RANK WORLD PORT OUT_DIR
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import gen  # noqa: E402
import dp_cases as C  # noqa: E402


def main():
    rank, world, port, out_dir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    import datetime
    import faulthandler
    import torch.distributed as dist
    faulthandler.dump_traceback_later(90, exit=True)          # a rank that waits for a dead peer says where
    from hhrl_vae_amd import dist as hdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    group = dist.group.WORLD
    res = {}
    C.run_vae(res, group, rank, world, hdist.shard_bounds)
    C.run_td3(res, "worker", group, rank, world, hdist.shard_bounds)
    C.run_td3(res, "triple", group, rank, world, hdist.shard_bounds)
    C.run_noise_streams(res, group)
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **res)
    dist.barrier()
    dist.destroy_process_group()
    faulthandler.cancel_dump_traceback_later()


if __name__ == "__main__":
    main()
