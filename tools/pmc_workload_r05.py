#!/usr/bin/env python
"""Fixed workload for rocprofv3 counter passes over the kernels added in round 5:
  InverseLBFGS m = 10 at n = 2^20 (persistent single-launch apply) x6, sorted opExtension / opRestriction 2e7 of 4e7
  (bit mask + ranks) x4 each, transposed dense block apply n = 16384, k = 8 (LDS-staged) x3, dense M*v n = 16384 f64
  (row bands) x4, ComplexF64 M*v n = 8192 (row bands) x4."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import __graft_entry__ as g

lo = g.load_package()
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(1)
n, m = 1 << 20, 10
op = lo.InverseLBFGSOperator(torch.float64, n, mem=m, device=dev)
for _ in range(m + 1):
    s = torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 2 - 1
    lo.push(op, s, (torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 1.5 + 0.5) * s)
x = torch.rand(n, dtype=torch.float64, device=dev, generator=gen)
out = torch.empty_like(x)
for _ in range(6):
    lo.mul(out, op, x, 1.0, 0.0)
torch.cuda.synchronize()
del op, x, out
nres, nidx = 40_000_000, 20_000_000
idx = (torch.randperm(nres, device=dev, generator=gen)[:nidx].sort().values + 1).cpu().numpy()
R = lo.opRestriction(idx, nres, device=dev)
u = torch.rand(nidx, dtype=torch.float64, device=dev, generator=gen)
full = torch.empty(nres, dtype=torch.float64, device=dev)
for _ in range(4):
    lo.mul(full, R.H, u)
for _ in range(4):
    lo.mul(u, R, full)
torch.cuda.synchronize()
del R, u, full
torch.cuda.empty_cache()
nn, k = 16384, 8
M = torch.rand(nn, nn, dtype=torch.float64, device=dev, generator=gen).t()
opM = lo.LinearOperatorFromMatrix(M)
V = torch.rand(k, nn, dtype=torch.float64, device=dev, generator=gen).t()
Rb = torch.empty(k, nn, dtype=torch.float64, device=dev).t()
for _ in range(3):
    lo.mul(Rb, opM.T, V)
torch.cuda.synchronize()
xv = torch.rand(nn, dtype=torch.float64, device=dev, generator=gen)
yv = torch.empty(nn, dtype=torch.float64, device=dev)
for _ in range(4):
    lo.mul(yv, opM, xv, 1.0, 0.0)                     # gemv_n_rows_kernel (64-row bands)
torch.cuda.synchronize()
del M, opM, V, Rb, xv, yv
torch.cuda.empty_cache()
nc = 8192
Mc = torch.complex(torch.rand(nc, nc, dtype=torch.float64, device=dev, generator=gen), torch.rand(nc, nc, dtype=torch.float64, device=dev, generator=gen)).t()
opC = lo.LinearOperatorFromMatrix(Mc)
xc = torch.complex(torch.rand(nc, dtype=torch.float64, device=dev, generator=gen), torch.rand(nc, dtype=torch.float64, device=dev, generator=gen))
yc = torch.empty(nc, dtype=torch.complex128, device=dev)
for _ in range(4):
    lo.mul(yc, opC, xc, 1.0, 0.0)                     # cgemv_rows_band_kernel
torch.cuda.synchronize()
print("pmc workload r05 done")
