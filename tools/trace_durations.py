#!/usr/bin/env python
"""Per-kernel durations out of a rocprofv3 --kernel-trace output directory: median / min per (kernel, grid) for kernels
whose name contains the given substring.   python tools/trace_durations.py <dir> <substring>"""
import collections
import csv
import os
import sys

root, pat = sys.argv[1], sys.argv[2]
found = False
for dp, _, fns in os.walk(root):
    for fn in fns:
        if not fn.endswith("kernel_trace.csv"):
            continue
        found = True
        d = collections.defaultdict(list)
        for r in csv.DictReader(open(os.path.join(dp, fn))):
            if pat in r["Kernel_Name"]:
                grid = r.get("Grid_Size_X") or r.get("Grid_Size") or "?"
                wg = r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or "?"
                d[(r["Kernel_Name"][:70], grid, wg)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for k, v in sorted(d.items(), key=lambda kv: (kv[0][0], int(kv[0][1]) if kv[0][1].isdigit() else 0)):
            v = sorted(v)
            print(f"{k[0]:70s} grid {k[1]:>9s} wg {k[2]:>4s} x{len(v):4d}  median {v[len(v)//2]/1e3:8.2f} us  min {v[0]/1e3:8.2f} us")
if not found:
    print("no *kernel_trace.csv under", root)
