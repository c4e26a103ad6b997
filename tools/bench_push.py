#!/usr/bin/env python
"""push!(op, s, y) at n = 5e7: the one-pass schedule (new pair held per lane, in-pass slot stores) beside
the copies + dual-x dots schedule it replaced (`mxlo_ctx_tune("push_fused", 0)`), wall time per push with a full memory
(host read of the accept / reject scalars included, as in an optimiser loop)."""
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
for kind, m in (("inv", 10), ("fwd", 20), ("fwd", 10), ("inv", 20)):
    make = {"inv": lo.InverseLBFGSOperator, "fwd": lo.LBFGSOperator}[kind]
    S = [torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 2 - 1 for _ in range(2)]
    Y = [(torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 1.5 + 0.5) * s for s in S]
    for fused in (2, 1, 0):
        ctx.tune("push_fused", int(fused > 0))
        ctx.tune("push_wide", int(fused == 2))
        op = make(torch.float64, n, mem=m, device=dev)
        for i in range(m + 2):
            lo.push(op, S[i % 2], Y[i % 2])
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            for i in range(4):
                lo.push(op, S[i % 2], Y[i % 2])
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 4 * 1e3)
        # necessary traffic of the one-pass schedule: (2m - 2) panel columns + s, y once per pass + the inserted columns
        passes = 2 * ((m + 9) // 10) if fused < 2 else 2 * (m // 20 + ((m % 20) + 9) // 10)
        cols = (2 * m - 2 + 2 * passes + (2 if kind == "inv" else 3))
        gb = cols * 8.0 * n / 1e9
        print(f"push! {kind} m={m:2d} n={n:.0e} {('one-pass wide', 'one-pass <=10 ', 'two-kernel   ')[2 - fused]}: {best:7.3f} ms"
              f"   ({gb:5.1f} GB necessary -> {gb / best:5.2f} TB/s = {gb / best / 8.0:.3f} of peak)", flush=True)
        del op
        torch.cuda.empty_cache()
    ctx.tune("push_fused", 1)
    ctx.tune("push_wide", 1)
    del S, Y
    torch.cuda.empty_cache()
