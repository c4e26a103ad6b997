#!/usr/bin/env python
"""Timeline out of a rocprofv3 --kernel-trace csv: for the LAST `count` kernels whose grid matches, print start offset,
duration and the gap to the previous kernel's end.   python tools/trace_timeline.py <dir> <count>"""
import csv
import os
import sys

root, count = sys.argv[1], int(sys.argv[2])
for dp, _, fns in os.walk(root):
    for fn in fns:
        if fn.endswith("kernel_trace.csv"):
            rows = sorted(csv.DictReader(open(os.path.join(dp, fn))), key=lambda r: int(r["Start_Timestamp"]))
            rows = rows[-count:]
            t0, prev = int(rows[0]["Start_Timestamp"]), None
            for r in rows:
                s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
                gap = "" if prev is None else f"gap {(s - prev)/1e3:6.2f}"
                name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("void mxlo::", "")[:48]
                grid = r.get("Grid_Size_X") or r.get("Grid_Size") or "?"
                print(f"{(s - t0)/1e3:9.2f} us  dur {(e - s)/1e3:7.2f}  {gap:12s} {name:48s} grid {grid}")
                prev = e
