#!/usr/bin/env python
"""opHouseholder at mid sizes (2^20 ... 2^23): the single-launch form with up to two workgroups per CU (tune key house_fused_per_cu,
round 6) against one per CU / the two-launch form; bit-exact comparison of the results of 1 vs 2 per CU where both apply, 1e-12 vs two launches."""
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


def timed(fn, reps=200):
    for _ in range(10):
        fn()
    best = 1e30
    for _ in range(3):
        tm.start()
        for _ in range(reps):
            fn()
        tm.stop()
        best = min(best, tm.elapsed_ms() / reps)
    return best * 1e3


for dt in (torch.float64, torch.float32):
    es = 8 if dt == torch.float64 else 4
    for n in (1 << 20, (1 << 21) - 3, 1 << 21, (1 << 21) + 7, 3_000_001, 1 << 22, (1 << 22) + 5, 1 << 23):
        h = torch.rand(n, dtype=dt, device=dev) - 0.5
        h /= torch.linalg.vector_norm(h)
        v, r0 = torch.rand(n, dtype=dt, device=dev) - 0.5, torch.rand(n, dtype=dt, device=dev)
        H = lo.opHouseholder(h)
        out, res = {}, r0.clone()
        for name, per_cu, fused in (("two per CU", 2, 1), ("one per CU", 1, 1), ("two launches", 2, 0)):
            ctx.tune("house_fused_per_cu", per_cu)
            ctx.tune("house_fused", fused)
            res.copy_(r0)
            lo.mul(res, H, v, 0.7, -1.3)
            out[name] = res.clone()
            us = timed(lambda: lo.mul(res, H, v, 1.0, 0.0))
            print(f"opHouseholder {str(dt)[6:]} n={n:9d} {name:13s}: {us:7.1f} us  ({5 * es * n / us / 1e3 / 8000:.3f} of peak on 40 B/elt)", flush=True)
        ctx.tune("house_fused_per_cu", 2)
        ctx.tune("house_fused", 1)
        e = float((out["two per CU"].double() - out["two launches"].double()).norm() / out["two launches"].double().norm())
        same = torch.equal(out["two per CU"], out["one per CU"])
        print(f"   two per CU vs two launches rel {e:.1e}; two per CU == one per CU bit for bit: {same}", flush=True)
        assert e <= (1e-12 if dt == torch.float64 else 2e-5)
