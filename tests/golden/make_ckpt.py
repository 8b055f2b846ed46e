"""Checkpoint interoperability fixtures: files written by the REFERENCE's own save() methods
(algorithm.py:129-132, 1368-1373) plus what the reference computes after load() -- one encode /
choose_action call and one more learn() step -- so that the tests can load reference checkpoints into
this package and continue training from them.

Run in the build container only (imports /root/reference/src/algorithm.py on CPU):
    python tests/golden/make_ckpt.py
Writes tests/golden/ckpt/*  (torch.save files of plain tensors, loadable with weights_only=True).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference/src")
import gen  # noqa: E402
import algorithm as REF  # noqa: E402  (the reference)
from make_golden import META, build_ref, load_np, put, sd_np  # noqa: E402

OUT = os.path.join(HERE, "ckpt")
torch.set_num_threads(1)


def vae():
    g, case = 4, "ckpt_vae_g4"
    enc = REF.Encoder(2, g)
    shapes = [(k, tuple(v.shape)) for k, v in enc.state_dict().items()]
    load_np(enc, gen.gen_state(shapes, gen.case_seed(case, "init")))
    for step in range(2):                                   # so that the optimizer has state to save
        torch.manual_seed(gen.case_seed(case, "eps%d" % step))
        enc.learn(gen.gen_lanes(8, 5, gen.case_seed(case, "x%d" % step)).tolist())
    prefix = os.path.join(OUT, "ref_encoder_dim_4")
    enc.save(prefix)
    out = {"meta": META}
    # a fresh reference object resumes from the files
    e2 = REF.Encoder(2, g)
    e2.load(prefix)
    lane = gen.gen_lanes(1, 9, gen.case_seed(case, "lane"))[0]
    out["encode/x"], out["encode/out"] = lane, e2.encode(lane.tolist())
    x = gen.gen_lanes(8, 5, gen.case_seed(case, "x_next"))
    seed = gen.case_seed(case, "eps_next")
    torch.manual_seed(seed)
    out["next/eps"] = torch.randn(8, g).numpy().copy()
    torch.manual_seed(seed)
    out["next/x"], out["next/loss"] = x, np.float64(e2.learn(x.tolist()))
    put(out, "next/param", sd_np(e2.state_dict()))
    np.savez_compressed(os.path.join(OUT, "ref_encoder_dim_4_expect.npz"), **out)
    print("wrote VAE checkpoint fixture")


def worker():
    case = "ckpt_worker"
    cfg = gen.cav_cfg(o=5, g=3, hidden=(16, 12), batch=8, cap=48, a_sched=1000, c_sched=1000)
    net = build_ref("worker", None, cfg)
    for a in ("actor", "critic"):
        mod = getattr(net, a)
        shp = [(k, tuple(v.shape)) for k, v in mod.state_dict().items()]
        st = gen.gen_state(shp, gen.case_seed(case, "init_" + a))
        load_np(mod, st)
        load_np(getattr(net, a + "_target"), st)
    rows = gen.gen_rows(40, net.memory.shape[1], gen.case_seed(case, "rows"))
    net.memory[:40] = rows
    net.pointer = 40
    for call in range(2):                                   # both optimizers get state
        np.random.seed(gen.case_seed(case, "call%d" % call))
        torch.manual_seed(gen.case_seed(case, "call%d" % call))
        net.learn()
    prefix = os.path.join(OUT, "ref_worker")
    net.save(prefix)
    out = {"meta": META, "rows": rows}
    n2 = build_ref("worker", None, cfg)
    n2.load(prefix)
    n2.memory[:40] = rows
    n2.pointer = 40
    o = gen.gen_rows(1, n2.obs_dim, gen.case_seed(case, "o"))[0]
    g_ = gen.gen_rows(1, n2.g_dim, gen.case_seed(case, "g"))[0]
    out["choose/o"], out["choose/g"], out["choose/out"] = o, g_, n2.choose_action(o, g_)
    # two more updates after the resume (the second one is a policy step for a fresh total_it)
    for call in range(2):
        seed = gen.case_seed(case, "next%d" % call)
        np.random.seed(seed)
        out["next%d/indices" % call] = np.random.choice(40, size=cfg["batch_size"])
        torch.manual_seed(seed)
        out["next%d/eps" % call] = torch.randn(cfg["batch_size"], 1).numpy().copy()
        np.random.seed(seed)
        torch.manual_seed(seed)
        n2.learn()
    put(out, "next/param/critic", sd_np(n2.critic.state_dict()))
    put(out, "next/param/actor", sd_np(n2.actor.state_dict()))
    put(out, "next/target/critic", sd_np(n2.critic_target.state_dict()))
    np.savez_compressed(os.path.join(OUT, "ref_worker_expect.npz"), **out)
    print("wrote WorkerTD3 checkpoint fixture")


def worker_decayed():
    """As worker(), with scheduler steps of 1 (actor) and 2 (critic): the saved optimizers carry an lr that
    StepLR has already halved, and the updates after the resume cross further decay boundaries
    (algorithm.py:1283-1294, 1375-1382)."""
    case = "ckpt_worker_decayed"
    cfg = gen.cav_cfg(o=5, g=3, hidden=(16, 12), batch=8, cap=48, a_sched=1, c_sched=2)
    net = build_ref("worker", None, cfg)
    for a in ("actor", "critic"):
        mod = getattr(net, a)
        shp = [(k, tuple(v.shape)) for k, v in mod.state_dict().items()]
        st = gen.gen_state(shp, gen.case_seed(case, "init_" + a))
        load_np(mod, st)
        load_np(getattr(net, a + "_target"), st)
    rows = gen.gen_rows(40, net.memory.shape[1], gen.case_seed(case, "rows"))
    net.memory[:40] = rows
    net.pointer = 40
    for call in range(4):
        np.random.seed(gen.case_seed(case, "call%d" % call))
        torch.manual_seed(gen.case_seed(case, "call%d" % call))
        net.learn()
    prefix = os.path.join(OUT, "ref_worker_decayed")
    net.save(prefix)
    out = {"meta": META, "rows": rows}
    out["saved/critic_lr"] = np.float64(net.critic_optimizer.param_groups[0]["lr"])
    out["saved/actor_lr"] = np.float64(net.actor_optimizer.param_groups[0]["lr"])
    n2 = build_ref("worker", None, cfg)
    n2.load(prefix)
    n2.memory[:40] = rows
    n2.pointer = 40
    n_next = 4
    lrs = np.zeros((n_next, 2))
    for call in range(n_next):
        seed = gen.case_seed(case, "next%d" % call)
        np.random.seed(seed)
        out["next%d/indices" % call] = np.random.choice(40, size=cfg["batch_size"])
        torch.manual_seed(seed)
        out["next%d/eps" % call] = torch.randn(cfg["batch_size"], 1).numpy().copy()
        np.random.seed(seed)
        torch.manual_seed(seed)
        n2.learn()
        lrs[call] = (n2.critic_optimizer.param_groups[0]["lr"], n2.actor_optimizer.param_groups[0]["lr"])
    out["next/lrs"] = lrs                                   # (critic, actor) lr after each resumed update
    put(out, "next/param/critic", sd_np(n2.critic.state_dict()))
    put(out, "next/param/actor", sd_np(n2.actor.state_dict()))
    put(out, "next/target/critic", sd_np(n2.critic_target.state_dict()))
    np.savez_compressed(os.path.join(OUT, "ref_worker_decayed_expect.npz"), **out)
    print("wrote WorkerTD3 decayed-lr checkpoint fixture: saved lrs %g %g, after resume %s"
          % (out["saved/critic_lr"], out["saved/actor_lr"], lrs.tolist()))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if "--decayed-only" not in sys.argv:
        vae()
        worker()
    worker_decayed()
