#!/usr/bin/env python3
"""The kernel sequence of the last two TD3 updates (a critic-only one and a policy step) of a kernel trace made
with tools/td3_trace.py, in launch order with each kernel's duration and the gap to its predecessor.

    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/td3_trace.py triple 64 40
    python tools/td3_sequence.py <dir>"""
import csv
import glob
import os
import sys

path = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# an update starts with the launch that brings the minibatch in
starts = [i for i, n in enumerate(names) if "gather_copy_group" in n or "gather_rows" in n]
if len(starts) < 3:
    starts = [i for i, n in enumerate(names) if "copy2d_group" in n]
lo, hi = starts[-3], starts[-1]
prev_end = None
for i in range(lo, hi):
    r = rows[i]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if i in starts:
        print("---- update")
    short = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("hhrl::", "")
    print("%7.2f us  gap %6.2f  grid %-14s wg %-5s %s" % ((e - s) / 1e3, 0.0 if prev_end is None else (s - prev_end) / 1e3,
          "%sx%sx%s" % (r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"), r.get("Grid_Size_Z", "?")),
          r.get("Workgroup_Size_X", "?"), short[:70]))
    prev_end = e
print("total %.1f us for two updates" % ((int(rows[hi - 1]["End_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / 1e3))
