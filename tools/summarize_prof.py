#!/usr/bin/env python3
"""Turns the raw output of tools/profile_bench.sh <tag> (gpurun_out/prof_<tag>/) into the summaries
kept under profiles/:

    <tag>_kernel_stats.csv          rocprofv3 --stats per-kernel totals / averages (names cut to 160 chars)
    <tag>_bench_under_rocprof.json  the bench line printed under the kernel-trace pass
    <tag>_hbm_traffic.json          HBM bytes per launch of the fused VAE kernels from the PMC passes:
                                    reads = 2 x FETCH_SIZE (gfx950 tallies 128-B requests at 64 B),
                                    writes = WRITE_SIZE, both reported by rocprofv3 in KiB

usage: python tools/summarize_prof.py <tag>        (on the CPU box, after the GPU call has merged its files)
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ("vae_lane_forward_kernel", "vae_car_fwd_bwd_kernel", "vae_lane_backward_kernel", "vae_reduce_grads_kernel")


def one(pattern):
    hits = glob.glob(pattern, recursive=True)
    if len(hits) != 1:
        raise SystemExit("expected one file for %s, found %d" % (pattern, len(hits)))
    return hits[0]


def counter_per_launch(path, counter):
    tot, cnt = {}, {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            for k in KERNELS:
                if k in row["Kernel_Name"]:
                    tot[k] = tot.get(k, 0.0) + float(row["Counter_Value"])
                    cnt[k] = cnt.get(k, 0) + 1
    return {k: tot[k] / cnt[k] for k in tot}


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    with open(one(src + "/trace/**/*_kernel_stats.csv"), newline="") as f, \
            open(os.path.join(dst, tag + "_kernel_stats.csv"), "w", newline="") as out:
        w = csv.writer(out)
        for row in csv.reader(f):
            row[0] = row[0][:160]
            w.writerow(row)
    shutil.copy(os.path.join(src, "bench_under_trace.json"), os.path.join(dst, tag + "_bench_under_rocprof.json"))
    fetch = counter_per_launch(one(src + "/fetch/**/*_counter_collection.csv"), "FETCH_SIZE")
    write = counter_per_launch(one(src + "/write/**/*_counter_collection.csv"), "WRITE_SIZE")
    with open(os.path.join(src, "bench_under_trace.json")) as f:
        workload = json.loads(f.read().strip().splitlines()[-1])["config"]["workload"]
    doc = {
        "workload": workload,
        "command": "tools/profile_bench.sh %s (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, one run each)" % tag,
        "correction": "read bytes = 2 x FETCH_SIZE (gfx950 tallies 128-B requests at 64 B); write bytes = WRITE_SIZE; "
                      "counters are reported in KiB",
        "kernels": {k: {"FETCH_SIZE_raw_KiB_per_launch": fetch[k], "WRITE_SIZE_raw_KiB_per_launch": write[k],
                        "hbm_bytes_per_launch": 1024.0 * (2.0 * fetch[k] + write[k])} for k in KERNELS if k in fetch},
    }
    with open(os.path.join(dst, tag + "_hbm_traffic.json"), "w") as f:
        json.dump(doc, f, indent=1)
        f.write("\n")
    for k, v in doc["kernels"].items():
        print("%-28s %10.0f B/launch" % (k, v["hbm_bytes_per_launch"]))


if __name__ == "__main__":
    main()
