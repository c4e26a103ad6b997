#!/usr/bin/env python
"""Traced workload for tools/pmc_sparse.sh: 10 applies of the sparse leaf on the 7-point Laplacian of a 160^3 grid."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import __graft_entry__ as g

lo = g.load_package()
dev = torch.device("cuda", 0)
gs = int(sys.argv[1]) if len(sys.argv) > 1 else 160
n = gs ** 3
i = torch.arange(n, device=dev)
z, y, xg = i // (gs * gs), (i // gs) % gs, i % gs
cols, rows = [], []
for dz, dy, dx in [(0, 0, 0), (1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]:
    ok = (z + dz >= 0) & (z + dz < gs) & (y + dy >= 0) & (y + dy < gs) & (xg + dx >= 0) & (xg + dx < gs)
    cols.append(i[ok]); rows.append((i + (dz * gs + dy) * gs + dx)[ok])
key = torch.unique(torch.cat(cols) * n + torch.cat(rows))
cols, rows = key // n, key % n
ccol = torch.zeros(n + 1, dtype=torch.int64, device=dev)
ccol[1:] = torch.cumsum(torch.bincount(cols, minlength=n), 0)
vals = torch.rand(key.numel(), dtype=torch.float64, device=dev) - 0.5
op = lo.LinearOperatorFromMatrix(torch.sparse_csc_tensor(ccol, rows, vals, size=(n, n)))
x, yv = torch.rand(n, dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.float64, device=dev)
for _ in range(10):
    lo.mul(yv, op, x, 1.0, 0.0)
torch.cuda.synchronize()
print("nnz", key.numel(), "n", n, "algorithmic bytes", key.numel() * 12 + n * 24)
