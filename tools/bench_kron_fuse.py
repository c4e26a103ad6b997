#!/usr/bin/env python
"""kron apply: the two dependent GEMMs as two launches (kron_fuse = 0) vs ONE launch with the XCD-local dependency
(kron_fuse = 1, gemm_glds.h: kron_fused_kernel) — time per apply (HIP events over a loop, and over a hipGraph replay of 50
applies: no host in the loop) and bit equality of the results. -> profiles/r06_kron_xcd_fuse.txt"""
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
PEAK = {torch.float64: 78.6, torch.float32: 157.3}


def timeit(fn, reps):
    for _ in range(5):
        fn()
    best = 1e9
    for _ in range(3):
        tm.start()
        for _ in range(reps):
            fn()
        tm.stop()
        best = min(best, tm.elapsed_ms() / reps)
    return best * 1e3


for dt in (torch.float64, torch.float32):
    for sz in (128, 256, 512, 768, 1024, 1536):
        gen = torch.Generator(device=dev).manual_seed(sz)
        A = (torch.rand(sz, sz, dtype=dt, device=dev, generator=gen) - 0.5).t()
        B = (torch.rand(sz, sz, dtype=dt, device=dev, generator=gen) - 0.5).t()
        x = torch.rand(sz * sz, dtype=dt, device=dev, generator=gen) - 0.5
        r0 = torch.rand(sz * sz, dtype=dt, device=dev, generator=gen)
        K = lo.kron(A, B)
        out = {}
        for fuse in (0, 1):
            ctx.tune("kron_fuse", fuse)
            res = r0.clone()
            lo.mul(res, K, x, 1.0, 0.0)
            resb = r0.clone()
            lo.mul(resb, K, x, 0.7, -1.3)
            rest = r0.clone()
            lo.mul(rest, K.T, x, 1.0, 0.0)
            torch.cuda.synchronize()
            us = timeit(lambda: lo.mul(res, K, x, 1.0, 0.0), 50)
            ust = timeit(lambda: lo.mul(rest, K.T, x, 1.0, 0.0), 50)
            out[fuse] = (us, ust, res.clone(), resb.clone(), rest.clone())
        ctx.tune("kron_fuse", 2)                      # timing experiment: the fused launch WITHOUT its wait (wrong results)
        us_nowait = timeit(lambda: lo.mul(res, K, x, 1.0, 0.0), 50)
        ctx.tune("kron_fuse", 0)
        same = all(torch.equal(out[0][k], out[1][k]) for k in (2, 3, 4))
        fl = 4.0 * sz ** 3
        print(f"kron {sz:4d}^2 (x) {sz:4d}^2 {str(dt)[6:]:8s} two launches {out[0][0]:7.1f} us ({fl / out[0][0] / 1e6 / PEAK[dt]:.3f})  "
              f"one launch {out[1][0]:7.1f} us ({fl / out[1][0] / 1e6 / PEAK[dt]:.3f})  | transpose {out[0][1]:7.1f} -> {out[1][1]:7.1f} us  | bits equal: {same} | one launch, wait removed (wrong results): {us_nowait:7.1f} us",
              flush=True)
