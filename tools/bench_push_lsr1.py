#!/usr/bin/env python
"""push!(op, s, y) of LSR1Operator at n = 5e7 with DISTINCT pairs (every push accepted, memory full): wall time per
push (host read of the accept / reject scalars included) beside the bytes the schedule has to move."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import __graft_entry__ as g

lo = g.load_package()
from linearoperators_jl_amd.device import get_ctx

dev = torch.device("cuda", 0)
ctx = get_ctx(dev)
gen = torch.Generator(device=dev).manual_seed(1)
n = int(os.environ.get("PUSH_N", 50_000_000))
reps = int(os.environ.get("PUSH_REPS", 6))
modes = [int(x) for x in os.environ.get("PUSH_MODES", "1,0").split(",")]
for m in (10, 5):
    npairs = m + 2 + reps
    S = [torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 2 - 1 for _ in range(npairs)]
    Y = [(torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 1.5 + 0.5) * s for s in S]
    for fused in modes:
        ctx.tune("push_fused", fused)
        for scaling in (True, False):
            op = lo.LSR1Operator(torch.float64, n, mem=m, scaling=scaling, device=dev)
            acc = 0
            for i in range(m + 2):
                lo.push(op, S[i], Y[i]); acc += op._last_push_accepted
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(reps):
                lo.push(op, S[m + 2 + i], Y[m + 2 + i]); acc += op._last_push_accepted
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
            # necessary bytes of a schedule that keeps the a_k panel: S and Y panels once for the Gram rows (2m), the a_k
            # panel once for y - B s (m), the rebuild A = [Y S] C (2m reads + m writes), s and y per pass, 2 inserted columns
            cols = 6 * m + 6
            gb = cols * 8.0 * n / 1e9
            print(f"push! lsr1 m={m:2d} scaling={int(scaling)} n={n:.0e} {'new' if fused else 'old'} schedule: {ms:7.3f} ms  accepted {acc}/{m + 2 + reps}"
                  f"   ({gb:5.1f} GB necessary -> {gb / ms:5.2f} TB/s = {gb / ms / 8.0:.3f} of peak)", flush=True)
            del op
            torch.cuda.empty_cache()
    ctx.tune("push_fused", 1)
    del S, Y
    torch.cuda.empty_cache()
