#!/usr/bin/env python
"""Differences of rocprofv3 HIP-API call counts between the runs of tools/contract_trace.sh."""
import csv
import os
import sys

d = sys.argv[1]


def counts(name):
    out = {}
    p = os.path.join(d, f"{name}_hip_api_stats.csv")
    if not os.path.exists(p):
        return None
    with open(p) as f:
        for row in csv.DictReader(f):
            out[row.get("Name") or row.get("name")] = int(row.get("Calls") or row.get("calls") or 0)
    return out


base, iters, pushes = counts("base"), counts("iters"), counts("pushes")
if not (base and iters and pushes):
    sys.exit("missing stats files")
names = sorted(set(base) | set(iters) | set(pushes))
print("rocprofv3 --hip-trace --stats, tools/contract_trace.py (n = 1,000,003 fp64, mem = 5)")
print("one warmed iteration = mul!(B) + mul!(Hinv) + mul!(H*D + B) + diag!(B) + solve_shifted_system!(B); per-call columns")
print("are (count in the longer run - count in the base run) / 1000\n")
print(f"{'HIP API':34s} {'base':>9s} {'+1000 iters':>12s} {'+1000 push!':>12s} {'per iteration':>14s} {'per push!':>10s}")
for nm in names:
    b, i, p = base.get(nm, 0), iters.get(nm, 0), pushes.get(nm, 0)
    print(f"{nm:34s} {b:9d} {i:12d} {p:12d} {(i - b) / 1000:14.3f} {(p - b) / 1000:10.3f}")
bad = [nm for nm in names if (iters.get(nm, 0) - base.get(nm, 0)) and not any(k in nm for k in ("Launch", "CallConfiguration", "GetLastError", "GetDevice", "SetDevice", "PeekAtLastError", "hipStreamGetCaptureInfo", "hipStreamIsCapturing"))]
print("\nAPIs whose count grows with warmed iterations, other than launches (hipLaunchKernel and its __hipPush/PopCall"
      "Configuration pair) and error / current-device queries:", bad or "none")
pb = {nm: (pushes.get(nm, 0) - base.get(nm, 0)) / 1000 for nm in names}
print(f"per push!: {pb.get('hipMemcpyAsync', 0):.0f} hipMemcpyAsync (2 device-to-device panel inserts + the ONE device-to-host copy of the "
      f"control scalars), {pb.get('hipStreamSynchronize', 0):.0f} hipStreamSynchronize (the wait for that copy), "
      f"{pb.get('hipMalloc', 0):.0f} hipMalloc, {pb.get('hipFree', 0):.0f} hipFree")
