#!/usr/bin/env python
"""Per-kernel averages of every counter found under a directory of rocprofv3 --pmc passes (counter_collection CSVs)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> values
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "?")
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    short = k if len(k) < 150 else k[:150] + "..."
    print(short)
    c = {name: sum(v) / len(v) for name, v in acc[k].items()}
    for name in sorted(c):
        print(f"    {name:36s} {c[name]:16.1f}   (n={len(acc[k][name])})")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c and c["SQ_BUSY_CYCLES"] > 0:
        print(f"    -> MFMA busy / SQ busy cycles        {c['SQ_VALU_MFMA_BUSY_CYCLES'] / c['SQ_BUSY_CYCLES']:.3f}")
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
        print(f"    -> L2 hit rate                       {c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']):.3f}")
    if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
        print(f"    -> LDS bank-conflict / LDS active    {c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']:.3f}")
