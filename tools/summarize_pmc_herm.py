#!/usr/bin/env python
"""opHermitian PMC passes -> HBM bytes per launch of the pass kernels next to the algorithmic bytes (strict lower triangle once).
python tools/summarize_pmc_herm.py <dir with pmc_* subdirectories>"""
import csv
import glob
import os
import sys
from collections import defaultdict

root, n = sys.argv[1], 16384
vals = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "herm" not in k:
            continue
        name = ("cherm_" if "cherm" in k else "herm_") + ("pass" if "pass" in k else "edge" if "edge" in k else "finish")
        vals[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name in sorted(vals):
    c = {k: sum(v) / len(v) for k, v in vals[name].items()}
    line = f"{name:14s} " + "  ".join(f"{k}={v:,.0f}" for k, v in sorted(c.items()))
    if "TCC_EA0_RDREQ_sum" in c:
        rd = (c["TCC_EA0_RDREQ_sum"] - c.get("TCC_EA0_RDREQ_32B_sum", 0.0)) * 128 + c.get("TCC_EA0_RDREQ_32B_sum", 0.0) * 32
        line += f"  -> read {rd/1e6:,.1f} MB"
        if name.endswith("pass"):
            alg = (16 if name.startswith("c") else 8) * n * (n - 1) / 2
            line += f" (algorithmic: strict lower triangle {alg/1e6:,.1f} MB; ratio {rd/alg:.4f})"
    if "TCC_EA0_WRREQ_sum" in c:
        wr = c.get("TCC_EA0_WRREQ_64B_sum", 0.0) * 64 + (c["TCC_EA0_WRREQ_sum"] - c.get("TCC_EA0_WRREQ_64B_sum", 0.0)) * 32
        line += f"  -> written {wr/1e6:,.2f} MB"
    print(line)
