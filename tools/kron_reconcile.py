#!/usr/bin/env python
"""VERDICT r2 #3a: the kron leg's HIP-event timing and its rocprofv3 kernel-trace durations IN THE SAME RUN.

  python tools/kron_reconcile.py [n]            -> prints the HIP-event time per apply (200 back-to-back applies)
  rocprofv3 --kernel-trace ... -- python tools/kron_reconcile.py   (tools/kron_reconcile.sh) and then
  python tools/kron_reconcile.py --summarise <dir>   -> per-kernel durations, gaps between consecutive kernels and the
                                                        first-start-to-last-end span per apply out of the trace
"""
import collections
import csv
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def summarise(root):
    for dp, _, fns in os.walk(root):
        for fn in fns:
            if not fn.endswith("kernel_trace.csv"):
                continue
            rows = [r for r in csv.DictReader(open(os.path.join(dp, fn))) if "gemm_glds_kernel" in r["Kernel_Name"]]
            rows.sort(key=lambda r: int(r["Start_Timestamp"]))
            if len(rows) < 40:
                continue
            rows = rows[-400:]                                     # the timed loop (the last 200 applies = 400 GEMMs)
            dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
            gap = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rows, rows[1:])]
            span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / (len(rows) / 2)
            med = lambda v: sorted(v)[len(v) // 2]
            print(f"kernel trace, last {len(rows)} gemm_glds_kernel dispatches (= {len(rows) // 2} applies):")
            print(f"  duration  median {med(dur) / 1e3:7.2f} us   min {min(dur) / 1e3:7.2f}   max {max(dur) / 1e3:7.2f}")
            print(f"  gap end->next start  median {med(gap) / 1e3:7.2f} us   min {min(gap) / 1e3:7.2f}   max {max(gap) / 1e3:7.2f}"
                  "   (negative: the next dispatch's start stamp precedes the previous end stamp)")
            print(f"  2 x median duration           = {2 * med(dur) / 1e3:7.2f} us per apply")
            print(f"  first start -> last end / applies = {span / 1e3:7.2f} us per apply (what a stream-ordered timer sees)")
            by = collections.Counter((r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Workgroup_Size_X") or r.get("Workgroup_Size")) for r in rows)
            print("  grids:", dict(by))
            return
    print("no gemm_glds_kernel rows under", root)


if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
    summarise(sys.argv[2])
    sys.exit(0)

import torch

import __graft_entry__ as g

lo = g.load_package()
from linearoperators_jl_amd.device import Timer, get_ctx

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda", 0)
ctx = get_ctx(dev)
tm = Timer(ctx)
gen = torch.Generator(device=dev).manual_seed(4)
A = ((torch.rand(n, n, dtype=torch.float64, device=dev, generator=gen) * 2 - 1) / 32).t()
B = ((torch.rand(n, n, dtype=torch.float64, device=dev, generator=gen) * 2 - 1) / 32).t()
K = lo.kron(A, B)
x = torch.rand(n * n, dtype=torch.float64, device=dev, generator=gen) * 2 - 1
res = torch.empty_like(x)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:                              # clocks up
    for _ in range(20):
        lo.mul(res, K, x, 1.0, 0.0)
    torch.cuda.synchronize()
for rep in range(3):
    tm.start()
    for _ in range(200):
        lo.mul(res, K, x, 1.0, 0.0)
    tm.stop()
    us = tm.elapsed_ms() / 200 * 1e3
    print(f"kron {n}^2 (x) {n}^2 f64: HIP events, 200 back-to-back applies: {us:7.2f} us per apply = "
          f"{4.0 * n ** 3 / us / 1e6:6.2f} TF ({4.0 * n ** 3 / us / 1e6 / 78.6:.3f} of 78.6 TF)", flush=True)
