#!/usr/bin/env python3
"""Per-kernel instruction statistics of the gfx950 code hipcc generates for csrc/*.hip: total
instructions, exec-mask branches, scalar branches, SGPR spill traffic (v_readlane/v_writelane),
global loads, waits, and the spill counts of the kernel descriptor.  Runs without a GPU.

    python tools/isa_stats.py [file.hip ...]       (default: every file in hhrl-vae_amd/csrc)"""
import glob
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-munsafe-fp-atomics", "--cuda-device-only", "-S"]


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "hhrl-vae_amd", "csrc", "*.hip")))
    for src in files:
        with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
            subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-o", tmp.name, src], check=True, stderr=subprocess.DEVNULL)
            lines = open(tmp.name).read().splitlines()
        spills = {}
        cur = None
        for l in lines:
            m = re.match(r"\s*\.name:\s+(\S+)", l)
            if m:
                cur = m.group(1)
            m = re.match(r"\s*\.(sgpr_spill_count|vgpr_spill_count|vgpr_count):\s+(\d+)", l)
            if m and cur:
                spills.setdefault(cur, {})[m.group(1)] = int(m.group(2))
        print(os.path.basename(src))
        for st, l in enumerate(lines):
            m = re.match(r"^(_Z\w+):", l)
            if not m:
                continue
            end = next((j for j in range(st, len(lines)) if lines[j].strip().startswith("s_endpgm")), None)
            if end is None:
                continue
            body = [x.strip() for x in lines[st + 1:end]]
            body = [x for x in body if x and x[0] not in ";." and not x.endswith(":")]
            c = Counter(x.split()[0] for x in body)
            sp = spills.get(m.group(1), {})
            name = re.sub(r"^_ZN4hhrl(2vf)?\d+", "", m.group(1))[:38]
            print("  %-38s instr %5d  exec-br %4d  s-br %4d  lane-spill %4d  gloads %4d  waits %4d  vgpr %3d  sspill %3d vspill %3d" % (
                name, len(body), c["s_cbranch_execz"] + c["s_cbranch_execnz"],
                c["s_cbranch_scc0"] + c["s_cbranch_scc1"] + c["s_cbranch_vccnz"] + c["s_cbranch_vccz"],
                c["v_readlane_b32"] + c["v_writelane_b32"], sum(v for k, v in c.items() if k.startswith("global_load")),
                c["s_waitcnt"], sp.get("vgpr_count", 0), sp.get("sgpr_spill_count", 0), sp.get("vgpr_spill_count", 0)))


if __name__ == "__main__":
    main()
