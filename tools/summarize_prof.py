#!/usr/bin/env python3
"""Turns the raw output of tools/profile_bench.sh <tag> (gpurun_out/prof_<tag>/) into the summaries
kept under profiles/:

    <tag>_kernel_stats.csv          rocprofv3 --stats per-kernel totals / averages (names cut to 160 chars)
    <tag>_bench_under_rocprof.json  the bench line printed under the kernel-trace pass
    <tag>_hbm_traffic.json          HBM bytes per launch of the fused VAE kernels from the PMC passes:
                                    reads = 2 x FETCH_SIZE (gfx950 tallies 128-B requests at 64 B),
                                    writes = WRITE_SIZE, both reported by rocprofv3 in KiB
    <tag>_mfma_util.json            matrix-core counters per launch of the same kernels (fourth pass, if present):
                                    SQ_VALU_MFMA_BUSY_CYCLES (cycles a SIMD's matrix pipe is busy, summed over SIMDs),
                                    SQ_INSTS_VALU_MFMA_MOPS_F32 (executed matrix FLOPs / 512), GRBM_GUI_ACTIVE

usage: python tools/summarize_prof.py <tag>        (on the CPU box, after the GPU call has merged its files)
       python tools/summarize_prof.py --td3 <tag>  (tools/profile_td3.sh <tag>: <tag>_td3_<config>_kernel_stats.csv,
                                                    each with the traced run's wall time per update in a comment line;
                                                    <tag>_td3_<config>_hbm_traffic.json and _mfma_util.json from the
                                                    counter passes, per kernel of the update and per update)
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ("vae_step_kernel", "vae_reduce_slabs_kernel", "vae_lane_forward_kernel", "vae_car_fwd_bwd_kernel",
           "vae_lane_backward_kernel", "vae_reduce_grads_kernel")


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


def short_name(full):
    """hhrl::mf::mlp3_forward_kernel<false>(...) -> mlp3_forward_kernel<false>"""
    n = full.split("(")[0].strip()
    if n.startswith("void "):
        n = n[5:]
    head, _, tail = n.partition("<")
    return head.split("::")[-1] + (("<" + tail) if tail else "")


def counters_by_kernel(path, names):
    """{kernel short name: {counter: (sum, dispatches)}} of one counter-collection CSV (our kernels only)."""
    out = {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if "hhrl::" not in row["Kernel_Name"] or row["Counter_Name"] not in names:
                continue
            d = out.setdefault(short_name(row["Kernel_Name"]), {})
            t, c = d.get(row["Counter_Name"], (0.0, 0))
            d[row["Counter_Name"]] = (t + float(row["Counter_Value"]), c + 1)
    return out


def td3_counters(d, cfg, tag, dst, avg_us, calls, updates):
    """The counter passes of one configuration -> <tag>_td3_<cfg>_hbm_traffic.json / _mfma_util.json."""
    fetch = glob.glob(d + "/fetch/**/*_counter_collection.csv", recursive=True)
    write = glob.glob(d + "/write/**/*_counter_collection.csv", recursive=True)
    mfma = glob.glob(d + "/mfma/**/*_counter_collection.csv", recursive=True)
    workload = "td3_" + cfg
    if len(fetch) == 1 and len(write) == 1:
        fe, wr = counters_by_kernel(fetch[0], ("FETCH_SIZE",)), counters_by_kernel(write[0], ("WRITE_SIZE",))
        doc = {"workload": workload,
               "command": "tools/profile_td3.sh %s (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, one run each, 40 updates after warm-up)" % tag,
               "correction": "read bytes = 2 x FETCH_SIZE (gfx950 tallies 128-B requests at 64 B); write bytes = WRITE_SIZE; "
                             "counters are reported in KiB", "kernels": {}}
        per_update = 0.0
        for k in sorted(fe):
            if k not in wr:
                continue
            f_avg = fe[k]["FETCH_SIZE"][0] / fe[k]["FETCH_SIZE"][1]
            w_avg = wr[k]["WRITE_SIZE"][0] / wr[k]["WRITE_SIZE"][1]
            b = 1024.0 * (2.0 * f_avg + w_avg)
            lpu = calls.get(k, 0) / max(updates, 1)
            doc["kernels"][k] = {"FETCH_SIZE_raw_KiB_per_launch": f_avg, "WRITE_SIZE_raw_KiB_per_launch": w_avg,
                                 "hbm_bytes_per_launch": b, "launches_per_update": lpu,
                                 "achieved_GBps": b / (avg_us[k] * 1e-6) / 1e9 if k in avg_us else None}
            per_update += b * lpu
        doc["hbm_bytes_per_update"] = per_update
        with open(os.path.join(dst, "%s_td3_%s_hbm_traffic.json" % (tag, cfg)), "w") as f:
            json.dump(doc, f, indent=1)
            f.write("\n")
        print("  %s: %.2f MB of HBM traffic per update" % (cfg, per_update / 1e6))
    if len(mfma) == 1:
        names = ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_BUSY_CYCLES", "SQ_WAVES", "GRBM_GUI_ACTIVE")
        c = counters_by_kernel(mfma[0], names)
        doc = {"workload": workload,
               "command": "tools/profile_td3.sh %s (rocprofv3 --pmc %s, 40 updates after warm-up)" % (tag, " ".join(names)),
               "definitions": "per launch; mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x the kernel-trace pass's average "
                              "duration x 2.4 GHz); executed_matrix_flops = 512 x SQ_INSTS_VALU_MFMA_MOPS_F32; update totals weight "
                              "every kernel by its launches per update in the kernel-trace pass", "kernels": {}}
        busy_cyc = flops = 0.0
        for k in sorted(c):
            if "SQ_VALU_MFMA_BUSY_CYCLES" not in c[k] or k not in avg_us:
                continue
            avg = lambda n: c[k][n][0] / c[k][n][1] if n in c[k] else None
            busy, mops = avg("SQ_VALU_MFMA_BUSY_CYCLES"), avg("SQ_INSTS_VALU_MFMA_MOPS_F32") or 0.0
            cyc = avg_us[k] * 1e-6 * CLOCK_GHZ * 1e9
            lpu = calls.get(k, 0) / max(updates, 1)
            doc["kernels"][k] = {"avg_launch_us": avg_us[k], "launches_per_update": lpu, "SQ_VALU_MFMA_BUSY_CYCLES": busy,
                                 "SQ_INSTS_VALU_MFMA_MOPS_F32": mops, "SQ_BUSY_CYCLES": avg("SQ_BUSY_CYCLES"), "SQ_WAVES": avg("SQ_WAVES"),
                                 "GRBM_GUI_ACTIVE": avg("GRBM_GUI_ACTIVE"), "mfma_busy_frac": busy / (N_SIMD * cyc),
                                 "executed_matrix_flops": 512.0 * mops, "executed_tflops": 512.0 * mops / (avg_us[k] * 1e-6) / 1e12}
            busy_cyc += busy * lpu
            flops += 512.0 * mops * lpu
        doc["executed_matrix_flops_per_update"] = flops
        doc["mfma_busy_simd_cycles_per_update"] = busy_cyc
        with open(os.path.join(dst, "%s_td3_%s_mfma_util.json" % (tag, cfg)), "w") as f:
            json.dump(doc, f, indent=1)
            f.write("\n")
        print("  %s: %.3f GFLOP executed on the matrix cores per update" % (cfg, flops / 1e9))


def td3(tag):
    dst = os.path.join(ROOT, "profiles")
    for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "prof_%s_td3_*" % tag))):
        cfg = os.path.basename(d)[len("prof_%s_td3_" % tag):]
        line = open(os.path.join(d, "line.json")).read().strip().splitlines()[-1]
        out_path = os.path.join(dst, "%s_td3_%s_kernel_stats.csv" % (tag, cfg))
        avg_us, calls = {}, {}
        with open(one(d + "/trace/**/*_kernel_stats.csv"), newline="") as f, open(out_path, "w", newline="") as out:
            out.write("# rocprofv3 --kernel-trace --stats -- python3 tools/td3_trace.py %s ; %s\n" % (cfg.replace("_b", " "), line))
            w = csv.writer(out)
            rows = list(csv.reader(f))
            for row in rows:
                if row and row[0] != "Name" and "hhrl::" in row[0]:
                    avg_us[short_name(row[0])] = float(row[3]) / 1e3
                    calls[short_name(row[0])] = int(row[1])
                row[0] = row[0][:160]
                w.writerow(row)
        print(out_path, line)
        info = json.loads(line)
        td3_counters(d, cfg, tag, dst, avg_us, calls, info["updates"] + info.get("warmup_updates", 0))


def main():
    if sys.argv[1] == "--td3":
        return td3(sys.argv[2])
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
    mfma = glob.glob(src + "/mfma/**/*_counter_collection.csv", recursive=True)
    if len(mfma) == 1:
        mfma_summary(mfma[0], os.path.join(dst, tag + "_kernel_stats.csv"), os.path.join(dst, tag + "_mfma_util.json"), tag, workload)


N_SIMD = 1024           # 256 CUs x 4
CLOCK_GHZ = 2.4         # MI355X peak engine clock: the clock the 157.3 TFLOP/s fp32 matrix peak is quoted at


def mfma_summary(path, stats_csv, out_path, tag, workload):
    avg_us = {}
    with open(stats_csv, newline="") as f:
        for row in csv.DictReader(f):
            for k in KERNELS:
                if k in row["Name"]:
                    avg_us[k] = float(row["AverageNs"]) / 1e3
    names = ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_BUSY_CYCLES", "SQ_WAVES", "GRBM_GUI_ACTIVE")
    c = {n: counter_per_launch(path, n) for n in names}
    doc = {"workload": workload,
           "command": "tools/profile_bench.sh %s, pass 4 (rocprofv3 --pmc %s)" % (tag, " ".join(names)),
           "definitions": "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles); kernel cycles = "
                          "the kernel-trace pass's average duration x 2.4 GHz (mfma_busy_frac) or GRBM_GUI_ACTIVE / 8 XCDs "
                          "(mfma_busy_frac_gui_active; reads high on dispatches this short); executed_matrix_flops = 512 x "
                          "SQ_INSTS_VALU_MFMA_MOPS_F32; executed_tflops = that / average duration",
           "kernels": {}}
    for k in KERNELS:
        if k not in c["SQ_VALU_MFMA_BUSY_CYCLES"] or k not in avg_us:
            continue
        busy, mops = c["SQ_VALU_MFMA_BUSY_CYCLES"][k], c["SQ_INSTS_VALU_MFMA_MOPS_F32"][k]
        gui = c["GRBM_GUI_ACTIVE"].get(k, 0.0)
        cyc = avg_us[k] * 1e-6 * CLOCK_GHZ * 1e9
        doc["kernels"][k] = {"avg_launch_us": avg_us[k], "SQ_VALU_MFMA_BUSY_CYCLES": busy, "SQ_INSTS_VALU_MFMA_MOPS_F32": mops,
                             "SQ_BUSY_CYCLES": c["SQ_BUSY_CYCLES"].get(k), "SQ_WAVES": c["SQ_WAVES"].get(k), "GRBM_GUI_ACTIVE": gui,
                             "mfma_busy_frac": busy / (N_SIMD * cyc),
                             "mfma_busy_frac_gui_active": busy / (N_SIMD * gui / 8.0) if gui else None,
                             "executed_matrix_flops": 512.0 * mops,
                             "executed_tflops": 512.0 * mops / (avg_us[k] * 1e-6) / 1e12}
    with open(out_path, "w") as f:
        json.dump(doc, f, indent=1)
        f.write("\n")
    for k, v in doc["kernels"].items():
        print("%-28s mfma busy %.3f  executed %.1f TFLOP/s" % (k, v["mfma_busy_frac"], v["executed_tflops"]))


if __name__ == "__main__":
    main()
