#!/usr/bin/env python
"""Small fixed workload for rocprofv3 counter passes: the BASELINE kernels at their bench sizes.
  householder mul! (n=1e8) x4, opDiagonal mul! x4, InverseLBFGS m=10 n=5e7 apply x2."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import __graft_entry__ as g

lo = g.load_package()
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(1)
n = 100_000_000
h = torch.rand(n, dtype=torch.float64, device=dev, generator=gen) - 0.5
h /= torch.linalg.vector_norm(h)
v = torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 2 - 1
res = torch.empty(n, dtype=torch.float64, device=dev)
H, D = lo.opHouseholder(h), lo.opDiagonal(h)
for _ in range(4):
    lo.mul(res, H, v, 1.0, 0.0)
for _ in range(4):
    lo.mul(res, D, v, 1.0, 0.0)
torch.cuda.synchronize()
del H, D, h, v, res
torch.cuda.empty_cache()
n, m = 50_000_000, 10
op = lo.InverseLBFGSOperator(torch.float64, n, mem=m, device=dev)
for _ in range(m):
    s = torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 2 - 1
    y = (torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 1.5 + 0.5) * s
    lo.push(op, s, y)
    del s, y
x = torch.rand(n, dtype=torch.float64, device=dev, generator=gen)
out = torch.empty_like(x)
for _ in range(2):
    lo.mul(out, op, x, 1.0, 0.0)
torch.cuda.synchronize()
print("pmc workload done")
