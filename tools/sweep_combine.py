#!/usr/bin/env python
"""Sweep panel_combine grid geometry on the L-BFGS workloads (run on the GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import __graft_entry__ as g

lo = g.load_package()
from linearoperators_jl_amd.device import Timer, get_ctx

dev = torch.device("cuda", 0)
ctx = get_ctx(dev)
tm = Timer(ctx)
gen = torch.Generator(device=dev).manual_seed(1)


def timeit(fn, reps=8):
    for _ in range(2):
        fn()
    tm.start()
    for _ in range(reps):
        fn()
    tm.stop()
    return tm.elapsed_ms() / reps


n = 50_000_000
for kind, m in (("inv", 10), ("fwd", 20)):
    op = {"inv": lo.InverseLBFGSOperator, "fwd": lo.LBFGSOperator}[kind](torch.float64, n, mem=m, device=dev)
    for _ in range(m):
        s = torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 2 - 1
        y = (torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 1.5 + 0.5) * s
        lo.push(op, s, y)
        del s, y
    x = torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 2 - 1
    out = torch.empty_like(x)
    for cb in (0, 4, 8, 16, 32, 64):
        ctx.tune("combine_blocks_per_cu", cb)
        ms = timeit(lambda: lo.mul(out, op, x, 1.0, 0.0))
        alg = (4 * m + 3) * 8.0 * n
        print(f"{kind} m={m} combine_blocks_per_cu={cb:2d}: {ms:7.3f} ms  {1e3/ms:6.1f} apply/s  {alg/ms/1e6:6.0f} GB/s", flush=True)
    ctx.tune("combine_blocks_per_cu", 16)
    del op, x, out
    torch.cuda.empty_cache()
