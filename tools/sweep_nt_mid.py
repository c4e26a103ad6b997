#!/usr/bin/env python
"""Mid sizes (operands between the 32 MiB of L2 and the 256 MiB Infinity Cache): does the second pass of a two-pass apply
find the first pass's lines in the Infinity Cache, and do nontemporal accesses (used above `nt_min_bytes`) help or hurt there?
Times opHouseholder and LBFGSOperator applies for n = 2^19 .. 2^24 with nt_min_bytes at 32 MiB (default), 256 MiB, 1 TiB."""
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
gen = torch.Generator(device=dev).manual_seed(3)


def rnd(n):
    return torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 2 - 1


def timeit(f, reps):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            f()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best * 1e6


NTS = ((32 << 20, "32MiB"), (256 << 20, "256MiB"), (1 << 40, "never"))
for lg in range(19, 25):
    n = 1 << lg
    h = rnd(n); h /= h.norm()
    v, res = rnd(n), torch.empty(n, dtype=torch.float64, device=dev)
    H = lo.opHouseholder(h)
    ops = {"Householder": (H, 40.0 * n)}
    for m in (5, 20):
        if 2 * m * 8 * n > 40e9:
            continue
        B = lo.LBFGSOperator(torch.float64, n, mem=m, device=dev)
        for _ in range(m + 1):
            s = rnd(n)
            lo.push(B, s, s * (torch.rand(n, dtype=torch.float64, device=dev, generator=gen) + 0.5))
        ops[f"LBFGS m={m}"] = (B, (4 * m + 3) * 8.0 * n)
    for name, (op, byts) in ops.items():
        row = []
        for nt, tag in NTS:
            ctx.tune("nt_min_bytes", nt)
            us = timeit(lambda: lo.mul(res, op, v, 1.0, 0.0), 50 if lg < 22 else 20)
            row.append(f"nt>{tag}: {us:8.1f} us ({byts / us / 1e3:6.0f} GB/s)")
        print(f"n=2^{lg} {name:12s} " + "  ".join(row), flush=True)
    ctx.tune("nt_min_bytes", 256 << 20)
    del ops, H
    torch.cuda.empty_cache()
