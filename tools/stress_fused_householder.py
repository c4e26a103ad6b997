#!/usr/bin/env python
"""Stress of the single-launch Householder's slot exchange (run under `timeout`): hundreds of thousands of back-to-back
launches with the grid size changing from call to call (1 ... 256 workgroups), interleaved with the two-launch path,
graph replays and unrelated streaming kernels; results are checked against a reference every few hundred calls."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import __graft_entry__ as g

lo = g.load_package()
from linearoperators_jl_amd.device import get_ctx

dev = torch.device("cuda", 0)
ctx = get_ctx(dev)
rng = np.random.default_rng(0)
sizes = [1, 100, 513, 2048, 4097, 30_000, 65_536, 131_072, 300_001, 524_288, 1_048_576, 2_097_159, 3_000_001, 4_194_304]   # the last three: two workgroups per CU (round 6)
ops = {}
for n in sizes:
    h = torch.rand(n, dtype=torch.float64, device=dev)
    h /= torch.linalg.vector_norm(h)
    v = torch.rand(n, dtype=torch.float64, device=dev)
    res = torch.empty(n, dtype=torch.float64, device=dev)
    H = lo.opHouseholder(h)
    want = v - 2.0 * torch.dot(h, v) * h
    ops[n] = (H, v, res, want, lo.capture_mul(torch.empty(n, dtype=torch.float64, device=dev), H, v, 1.0, 0.0))
big = torch.rand(20_000_000, dtype=torch.float64, device=dev)
D = lo.opDiagonal(big)
bres = torch.empty_like(big)
t0 = time.time()
calls = 0
total = int(os.environ.get("STRESS_CALLS", "300000"))
while calls < total:
    n = sizes[int(rng.integers(len(sizes)))]
    H, v, res, want, gcap = ops[n]
    k = int(rng.integers(1, 40))
    mode = int(rng.integers(10))
    if mode == 0:
        ctx.tune("house_fused", 0)
        for _ in range(k):
            lo.mul(res, H, v, 1.0, 0.0)
        ctx.tune("house_fused", 1)
    elif mode == 1:
        for _ in range(k):
            gcap.replay()
    elif mode == 2:
        lo.mul(bres, D, big, 1.0, 0.0)                      # a long streaming kernel right before the fused launches
        for _ in range(k):
            lo.mul(res, H, v, 1.0, 0.0)
    else:
        for _ in range(k):
            lo.mul(res, H, v, 1.0, 0.0)
    calls += k
    if rng.integers(8) == 0:
        torch.cuda.synchronize()
        err = (torch.linalg.vector_norm(res - want) / torch.linalg.vector_norm(want)).item()
        assert err <= 1e-12, (n, err, calls)
torch.cuda.synchronize()
print(f"{calls} launches in {time.time() - t0:.1f} s: no hang, every check within 1e-12", flush=True)
