#!/usr/bin/env python3
"""Headline benchmark: VAE training lane-samples/s (and agent-update
minibatches/s as extra fields) on N MI355X of one node.

    python bench.py --gpus N --steps K --warmup W

One "step" = one `Encoder.learn` on one resident batch (forward, ELBO,
backward, Adam) per GPU.  N=1 workload: 4096 lanes x 7 cars, latent 4, fp32,
synthetic lanes already in HBM.  N>1: every rank holds its own 4096-lane
shard (weak scaling, global batch N*4096), gradients are all-reduced (RCCL)
once per step.  Rank 0 prints ONE JSON line.

Started without a launcher (`python bench.py --gpus N`, no WORLD_SIZE in the
environment) the script starts its N ranks itself, as fresh child processes
of `python -m torch.distributed.run`, before anything in this process has
touched a GPU, and passes rank 0's JSON line on.  On a box with fewer than N
GPUs it prints a line that says so (value null) and exits 0; with
`--rehearse` it runs the N ranks on the one GPU over the gloo backend
instead -- every code path of the N-rank run, no scaling number.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from hhrl_vae_amd import work                      # noqa: E402  (shape arithmetic only; loads no GPU code)

# Chip peaks used for roofline fractions (MI355X, gfx950)
PEAK_FP32_MATRIX_TFLOPS = work.PEAK_FP32_MATRIX_TFLOPS     # dense f32 MFMA
PEAK_HBM_GBS = work.PEAK_HBM_GBS                           # HBM3E


def vae_flops_per_lane(N: int, g: int) -> float:
    """Algorithmic training FLOPs per lane-sample: forward MACs
    M = 12896*N + 16384 + 256*g (per-car layers x N, per-lane layers once),
    x 2 FLOP/MAC x 3 (forward, input gradient, weight gradient)."""
    return 6.0 * (12896.0 * N + 16384.0 + 256.0 * g)


def vae_executed_flops_per_lane(N: int, g: int) -> float:
    """Matrix-core FLOPs the fused kernels execute per lane-sample.  The first
    decoder layer over [d2 of the lane | loc embedding] is factored: per lane
    U = d2 A^T, dd2 = S A, dA = S^T d2 (3 x 4096 MACs) instead of 3 x 8192 per
    car; per car func.2 (3 x 4096 MACs) and the per-lane sums of df1 on the
    matrix core (16 x 128 MACs per car of a 16-car tile).  phi.0, loc, func.4
    and the position term of func.0 are element-wise."""
    lane = 3.0 * (512 + 4096 + 8192 + 128.0 * g + 128.0 * g + 4096) + 3.0 * 4096
    car = 3.0 * 4096 + 16.0 * 128
    return 2.0 * (lane + car * N)


def progress(msg: str) -> None:
    """One line on stderr from rank 0: a long run shows where it is (stdout carries the JSON line only)."""
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %6.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def synth_lanes(B: int, N: int, seed: int) -> np.ndarray:
    rs = np.random.RandomState(seed)
    x = rs.uniform(0.0, 1.0, size=(B, N, 1))
    v = rs.uniform(0.0, 1.08, size=(B, N, 1))
    return np.concatenate([x, v], axis=-1).astype(np.float32)


def cpu_baseline_vae(B: int, N: int, g: int, budget_s: float = 12.0) -> dict:
    """The CPU oracle (PyTorch fp32 restatement of the reference's
    Encoder.learn) timed on this box's host cores on the same batch shape."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import hhrl_oracle as O
    torch.manual_seed(0)
    shapes = O.vae_param_shapes(g)
    state = {}
    fan = 1
    for name, shp in shapes:
        if len(shp) == 2:
            fan = shp[1]
        state[name] = (torch.rand(shp) * 2 - 1) / fan ** 0.5
    vae = O.VAE(state, g)
    x = synth_lanes(B, N, 1)
    xl = x.tolist()                           # the reference's learn() receives a Python list
    eps = torch.randn(B, g)
    for _ in range(2):
        vae.learn(xl, eps)
    t0 = time.perf_counter()
    steps = 0
    while True:
        vae.learn(xl, eps)
        steps += 1
        if time.perf_counter() - t0 > budget_s and steps >= 5:
            break
    dt = time.perf_counter() - t0
    return {"value": B * steps / dt, "unit": "lane-samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d steps of the CPU oracle's Encoder.learn at B=%d N=%d g=%d (list->tensor included), %.1f s"
                      % (steps, B, N, g, dt),
            "ms_per_step": 1e3 * dt / steps}


AGENT_CASES = (("worker_td3_b64", "worker", 64, 4096), ("worker_td3_b4096", "worker", 4096, 8192),
               ("hatd3_triple_b64", "triple", 64, 2048), ("hatd3_triple_b1024", "triple", 1024, 2048))


def _agent_net(kind: str, batch: int, cap: int, device, group=None):
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import gen
    from hhrl_vae_amd.algorithm import HATD3Triple, WorkerTD3
    if kind == "worker":
        return WorkerTD3(gen.cav_cfg(cap=cap, batch=batch), device=str(device), group=group, seed=3407)
    return HATD3Triple(gen.light_cfg(cap=cap, batch=batch), device=str(device), group=group, seed=3407)


def cpu_baseline_td3(kind: str, batch: int, budget_s: float = 4.0) -> dict:
    """The CPU oracle's TD3 update (PyTorch fp32 restatement of the reference's learn()) timed on this box's
    host cores on the same minibatch shape, replay gather included, for about `budget_s` seconds."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import gen
    import hhrl_oracle as O
    cfg = gen.cav_cfg(batch=batch) if kind == "worker" else gen.light_cfg(batch=batch)
    fam = O.family_from_cfg(kind, cfg)
    t = lambda d: {k: torch.from_numpy(v) for k, v in d.items()}
    actors = [t(gen.gen_state(O.actor_param_shapes(fam, k), 21 + k)) for k in range(len(fam.a_dims))]
    net = O.TD3(fam, actors, t(gen.gen_state(O.critic_param_shapes(fam), 20)))
    rows = np.random.RandomState(3407).uniform(0, 1, size=(2048, fam.row_width))
    rs = np.random.RandomState(1)
    gen_t = torch.Generator().manual_seed(1)

    def one():
        idx = rs.randint(0, rows.shape[0], size=batch)
        noise = [torch.randn(batch, a, generator=gen_t) for a in fam.a_dims]
        net.learn(O.split_rows(fam, rows[idx]), noise, list(rs.permutation(len(fam.a_dims))))
    for _ in range(2):
        one()
    t0, n = time.perf_counter(), 0
    while True:
        one()
        n += 1
        if n >= 2 and n % 2 == 0 and time.perf_counter() - t0 > budget_s:       # whole policy cycles
            break
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "minibatches/s", "cores": torch.get_num_threads(), "kind": "port",
            "ms_per_update": 1e3 * dt / n,
            "sample": "%d updates of the CPU oracle's TD3 learn at batch %d (%s), %.1f s" % (n, batch, kind, dt)}


def agent_update_rates(device, steps: int, group=None, world: int = 1, cpu_baseline: bool = True,
                       graph_collectives=None, extras: bool = True) -> dict:
    """Extra fields: TD3 update minibatches/s from synthetic replay (WorkerTD3 row width 36 and HATD3Triple
    row width 1030), each with its place on the roofline (algorithmic FLOPs and bytes from the layer shapes,
    hhrl_vae_amd/work.py) and, on one GPU, the CPU oracle's rate on the same shape.  With a process group the
    batch named is PER RANK (weak scaling): a global minibatch of world x batch rows, one gradient all-reduce
    per optimizer step; time = max over ranks."""
    import torch.distributed as dist
    from hhrl_vae_amd import dist as hdist
    out = {}
    for name, kind, batch, n_rows in AGENT_CASES:
        progress("agent updates: %s%s" % (name, " x %d ranks" % world if group is not None else ""))
        net = _agent_net(kind, batch, n_rows, device, group)
        if graph_collectives is not None:
            net.graph_collectives = bool(graph_collectives)
        rank = dist.get_rank(group) if group is not None else 0
        rows = np.random.RandomState(3407 + rank).uniform(0, 1, size=(n_rows, net.fam.row_width))
        net.memory[:n_rows] = rows
        net.pointer = n_rows
        n_orders = 1
        for k in range(2, len(net.actors) + 1):
            n_orders *= k
        # warm up until every (policy step, actor order) launch chain has its final form (graph or eager);
        # a chain needs up to six runs, a given actor order comes up once in 2 * n_orders updates.  Data-parallel
        # ranks run the same number of updates (the actor order is rank 0's draw, so their chains agree).
        for it in range(60 * n_orders):
            net.learn()
            forms = net.launch_forms()
            done = it >= 30 and len(forms) == 1 + n_orders and "tuning" not in forms.values()
            if group is not None:
                done = hdist.reduce_scalar(1.0 if done else 0.0, "min", group, device) > 0.5
            if done:
                break
        # ... and until the device has been busy with this update for half a second: the update chains that
        # are bound by the GPU (device-resident replay) are measured at the clocks the chip settles on
        torch.cuda.synchronize()
        t_w, n_w = time.perf_counter(), 0
        n_warm = None
        if group is not None:
            n_warm = 200 if batch <= 64 else 50                                   # same count on every rank
        while (time.perf_counter() - t_w < 0.5) if n_warm is None else (n_w < n_warm):
            net.learn()
            n_w += 1
        torch.cuda.synchronize()
        n_timed = max(steps, int(0.25 * n_w / (time.perf_counter() - t_w)))          # at least a quarter of a second
        if group is not None:
            n_timed = int(hdist.reduce_scalar(float(n_timed), "max", group, device))
            torch.cuda.synchronize()
            dist.barrier(group=group)
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_timed):
            net.learn()
        torch.cuda.synchronize()
        if group is not None:
            dist.barrier(group=group)
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if group is not None:
            dt = hdist.reduce_scalar(dt, "max", group, device)
        w = work.td3_update(net.fam, batch)
        out[name] = {"minibatches_per_s": n_timed / dt, "ms_per_update": 1e3 * dt / n_timed,
                     "rows_per_s": world * n_timed * net.batch_size / dt, "updates_timed": n_timed,
                     "rows_per_rank": batch, "global_rows": world * batch,
                     # per GPU: every rank does this work on its shard
                     "roofline": dict(work.roofline(w["flops"], w["bytes"], dt / n_timed),
                                      flops_per_update=w["flops"], bytes_per_update=w["bytes"], macs_per_row=w["macs_per_row"]),
                     "launch_forms": sorted(set(net.launch_forms().values()))}
        prof_cfg = "%s_b%d" % (kind, batch)
        summary = td3_profile_summary(prof_cfg)
        if summary is not None:
            out[name]["roofline"].update(summary)
        if group is None and extras:
            # one policy evaluation as the agents call it every control step (host round trip included)
            fam = net.fam
            o = rows[0, :fam.streams[0].width]
            if fam.goal_dim:
                act = lambda: net.choose_action(o, rows[1, :fam.goal_dim])
            elif hasattr(net, "choose_time_action"):
                act = lambda: net.choose_time_action(o)
            else:
                act = lambda: net.choose_action(o)
            for _ in range(10):
                act()
            t0 = time.perf_counter()
            for _ in range(100):
                act()
            out[name]["choose_action_us"] = 1e4 * (time.perf_counter() - t0)
        del net
    if cpu_baseline and group is None:
        for name, kind, batch, _ in AGENT_CASES:
            progress("CPU oracle: " + name)
            out[name]["cpu_baseline"] = cpu_baseline_td3(kind, batch, 4.0 if batch <= 64 else 6.0)
    return out


def one_rank_dp_probe(device, steps: int, group, agents: dict, enc_ms: float, B: int, N: int, g: int) -> dict:
    """The data-parallel code path on ONE GPU: a one-rank RCCL group (the SUM is the identity, so the results are
    the single-device ones) makes every update take the N-rank route -- flat-gradient all-reduce through c10d
    between captured graph segments (or inside one graph: `graph_collective`), the separate optimizer launch
    of the VAE step.  `dp_overhead_us` = that route's time per update minus the single-device time measured in
    the same process: what the route itself costs before any inter-GPU traffic."""
    from hhrl_vae_amd.algorithm import Encoder
    out = {"backend": "nccl (RCCL), world size 1"}
    for tag, gc in (("segments", False), ("graph_collective", True)):
        progress("one-rank RCCL group, agent updates, collectives %s" % ("inside the graph" if gc else "between graph segments"))
        r = agent_update_rates(device, steps, group, 1, cpu_baseline=False, graph_collectives=gc, extras=False)
        for name, v in r.items():
            d = out.setdefault(name, {"ms_per_update_single_device": agents[name]["ms_per_update"]})
            d["ms_per_update_" + tag] = v["ms_per_update"]
            d["dp_overhead_us" + ("" if not gc else "_graph_collective")] = 1e3 * (v["ms_per_update"] - agents[name]["ms_per_update"])
            d["launch_forms_" + tag] = v["launch_forms"]
    progress("one-rank RCCL group, VAE step")
    enc = Encoder(2, g, device=str(device), seed=3407, group=group)
    x = torch.from_numpy(synth_lanes(B, N, 3407)).to(device)
    for _ in range(50):
        enc.learn(x, sync=False, global_count=B * N)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(400):
        enc.learn(x, sync=False, global_count=B * N)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / 400
    out["vae_step"] = {"ms_per_step_single_device": enc_ms, "ms_per_step_dp": ms, "dp_overhead_us": 1e3 * (ms - enc_ms)}
    return out


def inference_latency(enc, device) -> dict:
    """Extra fields: the per-call inference surface the agents use every simulated second
    (agent.py:326, 748, 809-811): one lane encoded per call as the reference does, eight lanes of one
    intersection in one launch, one decode call.  Host round trip included (the callers need numpy)."""
    rs = np.random.RandomState(7)
    lanes = [synth_lanes(1, int(n), 50 + i)[0] for i, n in enumerate(rs.randint(1, 30, size=8))]
    goal = rs.standard_normal(2 * enc.goal_dim).astype(np.float32) * 0.3
    out = {}

    def timed(fn, reps=200):
        # warm up by time, not by count: at least 50 ms of back-to-back calls, so that the figure is the call
        # at the clocks the chip settles on and not its first calls after an idle stretch
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < 0.05:
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return 1e6 * (time.perf_counter() - t0) / reps
    out["encode_one_lane_us"] = timed(lambda: enc.encode(lanes[0]))
    out["encode_8_lanes_one_call_us"] = timed(lambda: enc.encode_lanes(lanes))
    out["encode_8_lanes_8_calls_us"] = timed(lambda: [enc.encode(l) for l in lanes], reps=50)
    out["decode_one_us"] = timed(lambda: enc.decode(goal, [0.4]))
    return out


def scenario_rates(seconds: int = 60) -> dict:
    """Extra field: the four-intersection scenario (the reference's per-second call cadence, train.py:183-204, on a
    synthetic transition stream: 4 lights x 8 lanes, Encoder(2, 2), HATD3Triple, WorkerTD3; tools/four_intersections.py)
    with the reference's one-call-per-lane inference and with the batched calls.  Wall time of `seconds` simulated
    seconds after a 60-second warm-up that fills the replays and captures the graphs; `learning_path_wall_s` is the
    host time spent inside the classes (the rest is the synthetic stream itself)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    import four_intersections as fi
    import hhrl_vae_amd.algorithm as A
    out = {}
    for batched in (False, True):
        mcfg, wcfg = fi.make_cfgs(manager_capacity=640, worker_capacity=320)
        enc, mgr, wrk = A.Encoder(2, 2), A.HATD3Triple(mcfg), A.WorkerTD3(wcfg)
        st = fi.Stream()
        fi.run(enc, mgr, wrk, st, 60, batched=batched)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fi.run(enc, mgr, wrk, st, seconds, seed=1, batched=batched)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        path = sum(v for k, v in r.items() if k.startswith("wall_"))
        out["batched_calls" if batched else "one_call_per_lane"] = {
            "simulated_seconds": seconds, "wall_s": dt, "simulated_seconds_per_s": seconds / dt, "learning_path_wall_s": path,
            "manager_updates": r.get("learn_manager"), "worker_updates": r.get("learn_worker"),
            "encode_calls": r.get("encode"), "decode_calls": r.get("decode"),
            "wall_learn_manager_s": r.get("wall_learn_manager"), "wall_learn_worker_s": r.get("wall_learn_worker"),
            "wall_encode_s": r.get("wall_encode"), "wall_decode_s": r.get("wall_decode"), "wall_choose_s": r.get("wall_choose")}
    return out


def pmc_traffic(workload: str, kernel: str, suffix: str = "_hbm_traffic.json", field: str = "hbm_bytes_per_launch"):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary
    of this workload (rocprofv3 FETCH_SIZE / WRITE_SIZE passes, read bytes
    doubled as gfx950 requires); None when no summary covers the workload.
    With suffix "_mfma_util.json" / field "mfma_busy_frac": the matrix pipe's busy
    share of the launch from the SQ_VALU_MFMA_BUSY_CYCLES pass."""
    import glob
    import re

    def order(path):                       # r01_v8 before r01_v11: numbers compare as numbers
        return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", os.path.basename(path))]
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*" + suffix)), key=order):
        try:
            with open(path) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        k = d.get("kernels", {}).get(kernel)
        if d.get("workload") == workload and k is not None:
            best = {"bytes": k[field], "source": os.path.relpath(path, ROOT)}
    return best


def td3_profile_summary(cfg: str):
    """What the newest committed rocprofv3 summaries of an agent-update configuration (tools/profile_td3.sh ->
    profiles/<tag>_td3_<cfg>_*) say: the kernel with the largest share of the update, its matrix-pipe busy share,
    the HBM bytes one update moves and the matrix FLOPs it executes.  None when no summary covers the configuration."""
    import csv
    import glob
    import re

    def order(path):
        return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", os.path.basename(path))]

    def newest(suffix):
        hits = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_td3_%s_%s" % (cfg, suffix))), key=order)
        return hits[-1] if hits else None
    out = {}
    stats = newest("kernel_stats.csv")
    if stats is not None:
        rows = []
        with open(stats, newline="") as f:
            for row in csv.reader(l for l in f if not l.startswith("#")):
                if row and row[0] != "Name" and "hhrl::" in row[0]:
                    rows.append((int(row[2]), row[0], float(row[3]) / 1e3, float(row[4])))
        if rows:
            tot, name, avg_us, pct = max(rows)
            short = name.split("(")[0].replace("void ", "").split("::")[-1]
            out.update(kernel=short, kernel_avg_launch_us=avg_us, kernel_share_of_gpu_time=pct / 100.0,
                       kernel_stats_source=os.path.relpath(stats, ROOT))
    for suffix, pick in (("mfma_util.json", lambda d: {"executed_matrix_flops_per_update": d.get("executed_matrix_flops_per_update"),
                                                       "mfma_busy": (d["kernels"].get(out.get("kernel"), {}) or {}).get("mfma_busy_frac")}),
                         ("hbm_traffic.json", lambda d: {"traffic": d.get("hbm_bytes_per_update"), "traffic_unit": "bytes per update"})):
        path = newest(suffix)
        if path is None:
            continue
        try:
            with open(path) as f:
                d = json.load(f)
            out.update(pick(d))
            out[suffix.split(".")[0] + "_source"] = os.path.relpath(path, ROOT)
        except (OSError, ValueError, KeyError):
            pass
    return out or None


def _free_port() -> int:
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def launch_ranks(args, argv) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as children of
    `python -m torch.distributed.run` (this process never touches a GPU: counting devices does not
    initialise one) and pass rank 0's JSON line on."""
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus and not args.rehearse:
        msg = ("bench.py --gpus %d needs %d GPUs on this node, %d visible (add --rehearse to run the %d ranks on one "
               "GPU over gloo: all code paths, no scaling number)" % (args.gpus, args.gpus, n_dev, args.gpus))
        print(msg, file=sys.stderr)
        print(json.dumps({"metric": "VAE train lane-samples/s", "value": None, "unit": "lane-samples/s", "n_gpus": args.gpus,
                          "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "error": msg}))
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.rehearse:
        env["HHRL_BENCH_REHEARSE"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line)
    return proc.returncode if line is not None or proc.returncode != 0 else 1


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4096, help="lanes per GPU")
    ap.add_argument("--cars", type=int, default=7)
    ap.add_argument("--latent", type=int, default=4)
    ap.add_argument("--windows", type=int, default=9, help="further timed windows of --steps steps each (spread)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-agents", action="store_true")
    ap.add_argument("--no-dp-probe", action="store_true",
                    help="skip the one-rank RCCL pass that prices the data-parallel route (dp_one_rank in the line)")
    ap.add_argument("--rehearse", action="store_true",
                    help="N ranks on ONE GPU over gloo (exercises the N-rank path; not a scaling number)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args, sys.argv[1:]))

    # stdout carries the JSON line and nothing else: whatever a library prints there (RCCL announces its version on
    # the first communicator) is sent to stderr, the line itself goes to the descriptor stdout had
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    wd = float(os.environ.get("HHRL_BENCH_WATCHDOG", "0"))
    if wd > 0:                      # diagnostic: a rank still running after that many seconds says where and exits
        import faulthandler
        faulthandler.dump_traceback_later(wd, exit=True)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    rehearse = os.environ.get("HHRL_BENCH_REHEARSE") == "1" or (args.rehearse and world > 1)
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    group = None
    # HHRL_FORCE_DIST=1 exercises the data-parallel path (process group, flat all-reduce) with one rank
    if world > 1 or os.environ.get("HHRL_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
        group = dist.group.WORLD

    from hhrl_vae_amd import _lib
    from hhrl_vae_amd.algorithm import Encoder

    B, N, g = args.batch, args.cars, args.latent
    enc = Encoder(2, g, device=str(device), seed=3407, group=group)
    x = torch.from_numpy(synth_lanes(B, N, 3407 + rank)).to(device)
    gcount = B * N * world

    from hhrl_vae_amd import dist as hdist

    def barrier():
        torch.cuda.synchronize()
        if group is not None:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()

    def timed_window(k: int) -> float:
        """K steps between barriers; max over ranks."""
        barrier()
        t0 = time.perf_counter()
        for _ in range(k):
            enc.learn(x, sync=False, global_count=gcount)
        barrier()
        d = time.perf_counter() - t0
        return hdist.reduce_scalar(d, "max", group, device)

    progress("VAE step: %d lanes x %d cars, latent %d, %d rank(s)" % (B, N, g, world))
    # The chip leaves its idle clocks only after some tens of milliseconds of work, far more than --warmup steps of
    # 50 us each: a fixed, untimed stretch of steps first (the same count on every rank), so that the timed window
    # measures the step at the clocks a training run sees and not the ramp.  Then the --warmup steps, then the window.
    for _ in range(2000):
        enc.learn(x, sync=False, global_count=gcount)
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        enc.learn(x, sync=False, global_count=gcount)
    dt = timed_window(args.steps)                       # THE timed region: exactly --steps steps
    value = world * B * args.steps / dt
    # the same window repeated: how much one short window says
    more = sorted(timed_window(args.steps) for _ in range(max(args.windows, 0)))

    # the same region with the batch handed over from pinned host memory each step (reported beside `value`)
    x_host = x.cpu().pin_memory()
    for _ in range(5):
        x.copy_(x_host, non_blocking=True)
        enc.learn(x, sync=False, global_count=gcount)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x.copy_(x_host, non_blocking=True)
        enc.learn(x, sync=False, global_count=gcount)
    barrier()
    dt_h2d = time.perf_counter() - t0

    # per-kernel HIP-event timing over an identical region (events on the launch stream)
    lib = _lib.load()
    lib.hhrl_prof_reset()
    lib.hhrl_prof_enable(1)
    prof_steps = min(args.steps, 50)
    for _ in range(prof_steps):
        enc.learn(x, sync=False, global_count=gcount)
    torch.cuda.synchronize()
    lib.hhrl_prof_enable(0)
    prof = _lib.prof_report()
    lib.hhrl_prof_reset()

    agents = None
    if not args.no_agents:
        # every rank takes part in the data-parallel updates; only rank 0 reports
        agents = agent_update_rates(device, max(20, min(args.steps, 100)), group if world > 1 else None, world,
                                    cpu_baseline=not args.no_cpu_baseline)

    dp_probe = None
    if world == 1 and agents is not None and not args.no_dp_probe:
        # the N-rank route with one rank: never allowed to take the line down
        try:
            import torch.distributed as dist
            grp = group
            if grp is None:
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", str(_free_port()))
                dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
                grp = dist.group.WORLD
            dp_probe = one_rank_dp_probe(device, max(20, min(args.steps, 100)), grp, agents, 1e3 * dt / args.steps, B, N, g)
            if group is None:
                torch.cuda.synchronize()
                dist.destroy_process_group()
        except Exception as e:                                   # noqa: BLE001
            dp_probe = {"error": "%s: %s" % (type(e).__name__, e)}

    if rank == 0:
        kernels = {k: dict(v, ms_per_step=v["ms"] / prof_steps) for k, v in prof.items()}
        dom_name = max(kernels, key=lambda k: kernels[k]["ms"])
        dom = kernels[dom_name]
        mfma_bound = dom["flops"] > 0
        if mfma_bound:
            achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": dom_name, "achieved": achieved, "peak": PEAK_FP32_MATRIX_TFLOPS,
                    "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_MATRIX_TFLOPS, "traffic": None}
        else:
            achieved = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": achieved / PEAK_HBM_GBS, "traffic": None}
        workload = "vae_train_b%d_n%d_g%d" % (B, N, g)
        tr = pmc_traffic(workload, dom_name)
        if tr is not None:
            roof["traffic"] = tr["bytes"]
            roof["traffic_unit"] = "bytes per launch"
            roof["traffic_source"] = tr["source"]
        mb = pmc_traffic(workload, dom_name, "_mfma_util.json", "mfma_busy_frac")
        if mb is not None:
            roof["mfma_busy"] = mb["bytes"]
            roof["mfma_busy_source"] = mb["source"]
        # the same kernel priced by the matrix-core FLOPs it executes (see vae_executed_flops_per_lane)
        lane_macs = 512 + 4096 + 8192 + 128.0 * g + 128.0 * g + 4096
        executed = {"vae_car_fwd_bwd_kernel": 2.0 * (3 * 4096 + 16 * 128) * B * N,
                    "vae_lane_forward_kernel": 2.0 * (lane_macs + 4096) * B,
                    "vae_lane_backward_kernel": 2.0 * (2 * lane_macs + 2 * 4096) * B,
                    "vae_step_kernel": vae_executed_flops_per_lane(N, g) * B}.get(dom_name)
        if mfma_bound and executed is not None:
            ex = executed * dom["launches"] / (dom["ms"] * 1e-3) / 1e12
            roof["executed"] = ex
            roof["executed_frac"] = ex / PEAK_FP32_MATRIX_TFLOPS
        roof["avg_launch_us"] = 1e3 * dom["ms"] / max(dom["launches"], 1)
        roof["launches_per_step"] = dom["launches"] / prof_steps
        step_alg_tflops = vae_flops_per_lane(N, g) * B / (dt / args.steps) / 1e12
        gpu_ms_per_step = sum(v["ms"] for v in prof.values()) / prof_steps
        out = {
            "metric": "VAE train lane-samples/s", "value": value, "unit": "lane-samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "lanes_per_gpu": B, "cars_per_lane": N,
                       "latent": g, "global_batch": B * world, "parallelism": "dp%d" % world},
            "roofline": roof,
            "windows": {"n": len(more), "steps_each": args.steps,
                        "ms_per_step_min": 1e3 * more[0] / args.steps if more else None,
                        "ms_per_step_median": 1e3 * more[len(more) // 2] / args.steps if more else None,
                        "ms_per_step_max": 1e3 * more[-1] / args.steps if more else None},
            "value_with_h2d_per_step": world * B * args.steps / dt_h2d,
            "step_algorithmic_tflops": step_alg_tflops,
            "step_frac_of_fp32_matrix_peak": step_alg_tflops / PEAK_FP32_MATRIX_TFLOPS,
            "step_executed_matrix_tflops": vae_executed_flops_per_lane(N, g) * B / (dt / args.steps) / 1e12,
            "executed_over_algorithmic_flops": vae_executed_flops_per_lane(N, g) / vae_flops_per_lane(N, g),
            "gpu_kernel_ms_per_step": gpu_ms_per_step,
            "kernels": {k: {"launches_per_step": v["launches"] / prof_steps, "ms_per_step": v["ms_per_step"]}
                        for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["ms"])},
        }
        if rehearse:
            out["rehearsal"] = "%d ranks on one GPU over gloo: exercises the N-rank path, not a scaling number" % world
        if agents is not None:
            out["agent_updates"] = agents
        if dp_probe is not None:
            out["dp_one_rank"] = dp_probe
        if world == 1 and not args.no_agents:
            out["inference"] = inference_latency(enc, device)
            progress("four-intersection scenario")
            out["scenario"] = scenario_rates()
        if world == 1 and not args.no_cpu_baseline:
            progress("CPU oracle: VAE step")
            out["cpu_baseline"] = cpu_baseline_vae(B, N, g)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if group is not None:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
