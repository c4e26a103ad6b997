#!/usr/bin/env python
"""Summarise rocprofv3 outputs (kernel stats CSV + counter-collection CSVs) into a short text report."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    name = name.replace("mxlo::", "")
    for key in ("HouseholderOp", "DiagOp", "AxpbyOp", "ScaleOp", "FillOp", "CopyOp"):
        if key in name and "map_kernel" in name:
            return f"map_kernel<{key}>"
    if "panel_dots_kernel" in name:
        import re
        m = re.search(r"panel_dots_kernel<(\w+), (\d+), (\d+)", name)
        return f"panel_dots_kernel<{m.group(1)},VEC={m.group(2)},NC={m.group(3)}>" if m else "panel_dots_kernel"
    if "combine_kernel" in name:
        import re
        m = re.search(r"combine_kernel<(\w+), (\w+), (\d+)", name)
        return f"combine_kernel<MODE={m.group(3)}>" if m else "combine_kernel"
    return name[:70]


print("== kernel stats (rocprofv3 --kernel-trace --stats, bench.py --steps 20 --warmup 5) ==")
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))
    print(f"{'kernel':58s} {'calls':>7s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for r in rows[:25]:
        print(f"{short(r['Name']):58s} {r['Calls']:>7s} {float(r['AverageNs'])/1e3:10.1f} {float(r['MinNs'])/1e3:10.1f} "
              f"{float(r['MaxNs'])/1e3:10.1f} {float(r['Percentage']):6.2f}")

print()
print("== PMC passes (tools/pmc_workload.py): per-dispatch counter values, averaged per kernel ==")
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(f"-- {os.path.basename(d)}")
        for k, cs in agg.items():
            if not any(t in k for t in ("map_kernel", "panel_dots", "combine_kernel")):
                continue
            for cn, vals in cs.items():
                big = max(vals)
                sel = [v for v in vals if v > 0.5 * big] or vals   # the large (bench-size) dispatches
                print(f"   {k:52s} {cn:24s} n={len(sel):3d} avg={sum(sel)/len(sel):16.1f}")
