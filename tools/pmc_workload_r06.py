#!/usr/bin/env python
"""Fixed workload for rocprofv3 counter passes over the kernels added / changed in round 6:
  dense block apply M*V n = 16384, k = 8 (gemvb_n_rows_kernel) x3, kron 1024^2 f64 in one launch (kron_fused_kernel) x4,
  opHermitian n = 16384 (herm_pass_kernel, column-block strip order) x3 and its block form k = 4 x2."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import __graft_entry__ as g

lo = g.load_package()
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(1)
nn, k = 16384, 8
M = torch.rand(nn, nn, dtype=torch.float64, device=dev, generator=gen).t()
opM = lo.LinearOperatorFromMatrix(M)
V = torch.rand(k, nn, dtype=torch.float64, device=dev, generator=gen).t()
Rb = torch.empty(k, nn, dtype=torch.float64, device=dev).t()
for _ in range(3):
    lo.mul(Rb, opM, V)
torch.cuda.synchronize()
d = torch.rand(nn, dtype=torch.float64, device=dev, generator=gen)
H = lo.opHermitian(d, M)
x = torch.rand(nn, dtype=torch.float64, device=dev, generator=gen)
y = torch.empty_like(x)
for _ in range(3):
    lo.mul(y, H, x, 1.0, 0.0)
V4 = torch.rand(4, nn, dtype=torch.float64, device=dev, generator=gen).t()
R4 = torch.empty(4, nn, dtype=torch.float64, device=dev).t()
for _ in range(2):
    lo.mul(R4, H, V4)
torch.cuda.synchronize()
del M, opM, V, Rb, H, V4, R4
torch.cuda.empty_cache()
n = 1024
A = ((torch.rand(n, n, dtype=torch.float64, device=dev, generator=gen) * 2 - 1) / 32).t()
B = ((torch.rand(n, n, dtype=torch.float64, device=dev, generator=gen) * 2 - 1) / 32).t()
K = lo.kron(A, B)
xk = torch.rand(n * n, dtype=torch.float64, device=dev, generator=gen)
rk = torch.empty_like(xk)
for _ in range(4):
    lo.mul(rk, K, xk, 1.0, 0.0)
torch.cuda.synchronize()
print("pmc workload r06 done")
