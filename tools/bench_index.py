#!/usr/bin/env python
"""opRestriction / opExtension of a sorted index set: index-list kernels (mxlo_gather, mxlo_scatter_zero_sorted) against the
bit-mask + rank plan (mxlo_gather_plan, mxlo_scatter_zero_plan), fp64, several densities. Reports us, the rate of the
ALGORITHMIC bytes (24 B per index for the restriction, 16 B per index + 8 B per output for the extension) and
of the bytes each form has to MOVE (list form: the 32-byte sectors of v it touches; plan form: mask + ranks instead of
the index list)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import __graft_entry__ as g

lo = g.load_package()
from linearoperators_jl_amd import leaves
from linearoperators_jl_amd.device import Timer, get_ctx

dev = torch.device("cuda", 0)
ctx = get_ctx(dev)
tm = Timer(ctx)
gen = torch.Generator(device=dev).manual_seed(11)


def timeit(fn, reps=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    tm.start()
    for _ in range(reps):
        fn()
    tm.stop()
    return tm.elapsed_ms() / reps * 1e3


n = 40_000_000
full = torch.rand(n, dtype=torch.float64, device=dev, generator=gen)
print(f"# tools/bench_index.py: n = {n} fp64, sorted index sets; one MI355X, HIP events")
for dens in (0.5, 0.25, 0.1, 0.03):
    nidx = int(n * dens)
    idx = (torch.randperm(n, device=dev, generator=gen)[:nidx].sort().values + 1).cpu().numpy()
    sect = np.unique((idx - 1) // 4).size * 32.0          # 32-byte sectors of the long vector an index-driven gather touches
    u = torch.rand(nidx, dtype=torch.float64, device=dev, generator=gen)
    out = torch.empty(n, dtype=torch.float64, device=dev)
    for form in ("list", "plan"):
        leaves.PLAN_MIN_DENSITY = 1 if form == "plan" else 0
        leaves.PLAN_MIN_DENSITY_INV = leaves.PLAN_GATHER_DENSITY_INV = 1 << 30 if form == "plan" else 32
        R = lo.opRestriction(idx, n, device=dev)
        us_r = timeit(lambda: lo.mul(u, R, full))
        us_e = timeit(lambda: lo.mul(out, R.H, u))
        alg_r, alg_e = 24.0 * nidx, 16.0 * nidx + 8.0 * n
        mov_r = (16.0 * nidx + sect) if form == "list" else (8.0 * nidx + (8.0 * n if nidx * 8 >= n else sect) + n / 4.0)
        mov_e = (16.0 * nidx + 8.0 * n) if form == "list" else (8.0 * nidx + 8.0 * n + n / 4.0)
        print(f"density {dens:4.2f} {form}: restriction {us_r:7.1f} us  alg {alg_r / us_r / 1e6:5.2f} TB/s ({alg_r / us_r / 8e6:.2f})  moved {mov_r / 1e6:6.0f} MB {mov_r / us_r / 1e6:5.2f} TB/s ({mov_r / us_r / 8e6:.2f})"
              f" | extension {us_e:7.1f} us  alg {alg_e / us_e / 1e6:5.2f} TB/s ({alg_e / us_e / 8e6:.2f})  moved {mov_e / 1e6:6.0f} MB {mov_e / us_e / 1e6:5.2f} TB/s ({mov_e / us_e / 8e6:.2f})", flush=True)
        del R
