"""The product classes with a REAL process group of two: `Encoder.learn(group=...)`, `WorkerTD3` and
`HATD3Triple` run as two freshly started child processes that share the one GPU and exchange gradients over
the gloo backend (tests/dp_worker.py), each on its shard of a global minibatch -- uneven shards included -- and
must land on what one process computes on the whole batch: the all-reduced gradients, the loss, the learning
rates and the parameters.  The TD3 classes pass through eager runs, the capture of one HIP graph per stretch
between two all-reduces, and replays."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

import dp_cases as C
from golden_util import adam_path_close, assert_close, sub
from test_gpu_kernels import scaled_close

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope="module")
def ranks(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("dp2"))
    port = str(_free_port())
    env = dict(os.environ, OMP_NUM_THREADS="2")
    # each rank's output goes to a file of its own (no pipe that could fill while the parent waits for the other rank)
    log_paths = [os.path.join(out, "rank%d.log" % r) for r in range(2)]
    files = [open(lp, "w") for lp in log_paths]
    procs = [subprocess.Popen([sys.executable, "-u", os.path.join(HERE, "dp_worker.py"), str(r), "2", port, out],
                              stdout=files[r], stderr=subprocess.STDOUT, env=env) for r in range(2)]
    try:
        for p in procs:
            p.wait(timeout=150)
    except subprocess.TimeoutExpired:
        for q in procs:
            q.kill()
    for f in files:
        f.close()
    logs = [open(lp).read() for lp in log_paths]
    bad = [r for r, p in enumerate(procs) if p.returncode != 0]
    assert not bad, "\n".join("rank %d ended with %s:\n%s" % (r, procs[r].returncode, logs[r][-5000:]) for r in range(2))
    return [dict(np.load(os.path.join(out, "rank%d.npz" % r))) for r in range(2)]


@pytest.fixture(scope="module")
def single():
    res = {}
    C.run_vae(res, None, 0, 1, None)
    C.run_td3(res, "worker", None, 0, 1, None)
    C.run_td3(res, "triple", None, 0, 1, None)
    torch.cuda.synchronize()
    return res


def test_replicas_stay_identical(ranks):
    r0, r1 = ranks
    for k in r0:
        if not k.startswith("noise/"):
            assert np.array_equal(r0[k], r1[k]), k


def test_ranks_share_weights_but_not_noise(ranks):
    r0, r1 = ranks
    assert np.array_equal(r0["noise/vae_w"], r1["noise/vae_w"]) and np.array_equal(r0["noise/td3_w"], r1["noise/td3_w"])
    assert not np.array_equal(r0["noise/vae_eps"], r1["noise/vae_eps"])
    assert not np.array_equal(r0["noise/td3_eps"], r1["noise/td3_eps"])


@pytest.mark.parametrize("tag", [c[0] for c in C.VAE_CASES])
def test_vae_two_ranks_match_one_process(ranks, single, tag):
    r0, pre = ranks[0], "vae_%s" % tag
    assert_close("losses", r0[pre + "/losses"], single[pre + "/losses"], rtol=1e-5, atol=0)
    g1 = sub(single, pre + "/grad0")
    for k, w in g1.items():
        scaled_close("grad0/" + k, r0[pre + "/grad0/" + k], w, rtol=1e-4, atol_rel=2e-6)
    adam_path_close("param0", sub(r0, pre + "/param0"), sub(single, pre + "/param0"), [g1], [1e-3])
    # two more steps on top: the first step's gradient bounds how far a near-zero element may have drifted
    adam_path_close("param", sub(r0, pre + "/param"), sub(single, pre + "/param"), [g1] * 3, [1e-3] * 3, g_floor=5e-6)


@pytest.mark.parametrize("kind", ["worker", "triple"])
def test_td3_two_ranks_match_one_process(ranks, single, kind):
    r0 = ranks[0]
    assert list(r0[kind + "/forms"]) == ["graph"], r0[kind + "/forms"]       # every launch chain ended up captured
    nets = sorted({k.split("/")[2] for k in single if k.startswith(kind + "/param/")})
    grads = {n: [] for n in nets}
    lrs = {n: [] for n in nets}
    for it in range(C.TD3_UPDATES):
        pre = "%s/it%d" % (kind, it)
        assert r0[pre + "/lr_critic"] == single[pre + "/lr_critic"]
        for n in nets:
            g1 = sub(single, pre + "/grad/" + n)
            if not g1:
                continue
            for k, w in g1.items():
                scaled_close("%s/grad/%s/%s" % (pre, n, k), r0["%s/grad/%s/%s" % (pre, n, k)], w, rtol=1e-4, atol_rel=1e-5)
            grads[n].append(g1)
            lrs[n].append(1e-3 if n == "critic" else 1e-4)       # upper bounds: the schedulers only ever halve them
    for n in nets:
        adam_path_close(kind + "/param/" + n, sub(r0, "%s/param/%s" % (kind, n)), sub(single, "%s/param/%s" % (kind, n)),
                        grads[n], lrs[n], g_floor=1e-5)
        adam_path_close(kind + "/target/" + n, sub(r0, "%s/target/%s" % (kind, n)), sub(single, "%s/target/%s" % (kind, n)),
                        grads[n], lrs[n], g_floor=1e-5)
