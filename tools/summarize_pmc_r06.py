#!/usr/bin/env python
"""Summarise tools/pmc_r06.sh: per kernel of interest the average duration and the memory-side traffic per launch
(reads: TCC_EA0_RDREQ x 128 B minus the 32-byte ones counted at 32 B — the guide's gfx950 unit; writes: WRREQ_64B x 64 B
+ the rest x 32 B), next to the bytes the design says the launch has to move."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
WANT = {"gemvb_n_rows_kernel": ("dense M*V n=16384 k=8 f64, row bands + V in LDS", 8.0 * 16384 * 16384 + 2 * 8.0 * 16384 * 8),
        "kron_fused_kernel": ("kron 1024^2 (x) 1024^2 f64, both GEMMs in one launch", 8.0 * 1024 * 1024 * (2 + 1 + 1 + 2)),
        "herm_pass_block_kernel": ("opHermitian n=16384 block k=4, pass (triangle once)", 4.0 * 16384 * 16384),
        "herm_pass_kernel": ("opHermitian n=16384, pass (column-block strip order)", 4.0 * 16384 * 16384)}


def key(name):
    for k in WANT:
        if k in name:
            return k
    return None


dur = {}
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = key(r["Name"])
        if k:
            dur[k] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
ctr = defaultdict(lambda: defaultdict(list))
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if os.path.isdir(d):
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = key(r["Kernel_Name"])
                if k:
                    ctr[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# tools/pmc_r06.sh: memory-side traffic per launch of the round-5 kernels (one MI355X; rocprofv3 --pmc, separate passes)")
for k, (what, design) in WANT.items():
    c = {n: sum(v) / len(v) for n, v in ctr[k].items()}
    if not c:
        print(f"{k}: no counters collected")
        continue
    rd = (c.get("TCC_EA0_RDREQ_sum", 0) - c.get("TCC_EA0_RDREQ_32B_sum", 0)) * 128 + c.get("TCC_EA0_RDREQ_32B_sum", 0) * 32
    wr = c.get("TCC_EA0_WRREQ_64B_sum", 0) * 64 + (c.get("TCC_EA0_WRREQ_sum", 0) - c.get("TCC_EA0_WRREQ_64B_sum", 0)) * 32
    calls, us = dur.get(k, (0, float("nan")))
    hit, miss = c.get("TCC_HIT_sum", 0), c.get("TCC_MISS_sum", 0)
    print(f"{k:26s} {what}")
    print(f"    avg launch {us:8.1f} us ({calls} calls)   read {rd / 1e6:8.1f} MB  written {wr / 1e6:8.1f} MB  total {(rd + wr) / 1e6:8.1f} MB"
          f" = {(rd + wr) / design:5.3f} x the {design / 1e6:.1f} MB of the design   -> {(rd + wr) / us / 1e6:5.2f} TB/s at the memory side;"
          f" L2 hit rate {hit / max(hit + miss, 1):.3f}")
