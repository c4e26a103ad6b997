#!/usr/bin/env python
"""Sorted index extension: sweep of the tiles-per-workgroup knob (tune key extend_tiles_per_block)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import __graft_entry__ as g

lo = g.load_package()
from linearoperators_jl_amd.device import Timer, get_ctx

dev = torch.device("cuda", 0)
ctx = get_ctx(dev)
tm = Timer(ctx)
gen = torch.Generator(device=dev).manual_seed(1)
n, nidx = 100_000_000, 50_000_000


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    tm.start()
    for _ in range(reps):
        fn()
    tm.stop()
    return tm.elapsed_ms() / reps


for dt, es in ((torch.float64, 8), (torch.float32, 4)):
    res = torch.empty(n, dtype=dt, device=dev)
    out = torch.rand(nidx, dtype=dt, device=dev, generator=gen)
    for nm, idx in (("sorted+dups", torch.sort(torch.randint(1, n + 1, (nidx,), device=dev, generator=gen)).values.cpu().numpy()),
                    ("sorted unique", (torch.randperm(n, device=dev, generator=gen)[:nidx].sort().values + 1).cpu().numpy()),
                    ("random", torch.randint(1, n + 1, (nidx,), device=dev, generator=gen).cpu().numpy())):
        nu = np.unique(idx).size
        P = lo.opRestriction(idx, n, device=dev)
        for tpb in (0, 1, 2, 4, 8, 16):
            ctx.tune("extend_tiles_per_block", tpb)
            ms = timeit(lambda: lo.mul(res, P.H, out))
            nb = (8 + es + (0 if nm == 'sorted unique' else 8)) * nu + es * n      # idx + u (+ pos) per surviving entry + res once
            print(f"{dt} {nm:14s} tiles/wg={tpb:3d}: {ms*1e3:8.1f} us  {nb/ms/1e6:6.0f} GB/s  {nb/ms/1e6/8000:5.3f}", flush=True)
        del P
ctx.tune("extend_tiles_per_block", 0)
