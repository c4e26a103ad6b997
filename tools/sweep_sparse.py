#!/usr/bin/env python
"""Sparse apply (mxlo_csc_mul) on four patterns (the harness the round-4 kernel variants were compared with; the variants
themselves are described in csrc/sparse_kernels.h) -> profiles/r04_sweep_sparse.txt"""
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


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    tm.start()
    for _ in range(reps):
        fn()
    tm.stop()
    return tm.elapsed_ms() / reps


def stencil_csc(gs, offsets):
    n = gs ** 3
    i = torch.arange(n, device=dev)
    z, y, xg = i // (gs * gs), (i // gs) % gs, i % gs
    cols, rows = [], []
    for dz, dy, dx in offsets:
        ok = (z + dz >= 0) & (z + dz < gs) & (y + dy >= 0) & (y + dy < gs) & (xg + dx >= 0) & (xg + dx < gs)
        cols.append(i[ok]); rows.append((i + (dz * gs + dy) * gs + dx)[ok])
    key = torch.unique(torch.cat(cols) * n + torch.cat(rows))
    cols, rows = key // n, key % n
    ccol = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    ccol[1:] = torch.cumsum(torch.bincount(cols, minlength=n), 0)
    vals = torch.rand(key.numel(), dtype=torch.float64, device=dev, generator=gen) - 0.5
    return torch.sparse_csc_tensor(ccol, rows, vals, size=(n, n))


def banded_csc(n, per_row, spread):
    cols = torch.arange(n, device=dev).repeat_interleave(per_row)
    offs = torch.stack([torch.randperm(2 * spread + 1, device=dev, generator=gen)[:per_row] for _ in range(64)])
    off = offs[torch.randint(0, 64, (n,), device=dev, generator=gen)].reshape(-1) - spread
    key = torch.unique(cols * n + (cols + off).clamp_(0, n - 1))
    cols, rows = key // n, key % n
    ccol = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    ccol[1:] = torch.cumsum(torch.bincount(cols, minlength=n), 0)
    vals = torch.rand(key.numel(), dtype=torch.float64, device=dev, generator=gen) - 0.5
    return torch.sparse_csc_tensor(ccol, rows, vals, size=(n, n))


seven = [(0, 0, 0), (1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]
full27 = [(a, b, c) for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1)]
cases = [("7-point Laplacian 160^3", stencil_csc(160, seven)), ("27-point stencil 128^3", stencil_csc(128, full27)),
         ("random band 16/row +-2000 n=4e6", banded_csc(4_000_000, 16, 2000)),
         ("random columns 13.6/row over +-1.5e6 n=4e6", banded_csc(4_000_000, 16, 1_500_000))]
for name, M in cases:
    op = lo.LinearOperatorFromMatrix(M)
    n, nnz = M.shape[0], M.values().numel()
    x, y = torch.rand(n, dtype=torch.float64, device=dev, generator=gen), torch.empty(n, dtype=torch.float64, device=dev)
    nbytes = nnz * 12 + n * 24
    for xcds in (1, 8):
        ctx.tune("sp_xcds", xcds)
        out = []
        for o in (op, lo.transpose(op)):
            ms = timeit(lambda: lo.mul(y, o, x, 1.0, 0.0))
            out.append(f"{ms * 1e3:7.1f} us {nbytes / ms / 1e6 / 8000:5.3f}")
        print(f"{name:44s} sp_xcds {xcds}:  A*x {out[0]}   A'*x {out[1]}", flush=True)
ctx.tune("sp_xcds", 8)

# block apply: A read once per 8 columns (mxlo_csc_mul_block) vs k single-vector applies
name, M = cases[0]
op = lo.LinearOperatorFromMatrix(M)
n, nnz = M.shape[0], M.values().numel()
for k in (2, 4, 8, 16):
    V = torch.rand(k, n, dtype=torch.float64, device=dev, generator=gen).t()
    R = torch.empty(k, n, dtype=torch.float64, device=dev).t()
    ms_b = timeit(lambda: lo.mul(R, op, V, 1.0, 0.0), 10)
    def cols():
        for j in range(k):
            lo.mul(R[:, j], op, V[:, j], 1.0, 0.0)
    ms_c = timeit(cols, 10)
    nb = nnz * 12 * ((k + 7) // 8) + n * 8 * ((k + 7) // 8) + 2 * n * 8 * k
    print(f"{name} on an n x {k:2d} block: {ms_b * 1e3:8.1f} us ({nb / ms_b / 1e6 / 8000:5.3f} of HBM peak on A once per 8 columns + the block)"
          f"   vs {k} applies {ms_c * 1e3:8.1f} us  (x{ms_c / ms_b:4.2f})", flush=True)

# complex element types: the native instantiation (mxlo_csc_mul_c) vs the real-planes form (four real sweeps + split / join)
Mr = cases[0][1]
Mc = torch.sparse_csc_tensor(Mr.ccol_indices(), Mr.row_indices(),
                             torch.complex(Mr.values(), torch.rand(Mr.values().numel(), dtype=torch.float64, device=dev, generator=gen) - 0.5),
                             size=Mr.shape)
n, nnz = Mc.shape[0], Mc.values().numel()
xc = torch.complex(torch.rand(n, dtype=torch.float64, device=dev, generator=gen), torch.rand(n, dtype=torch.float64, device=dev, generator=gen))
yc = torch.empty_like(xc)
nbc = nnz * 20 + n * 8 + 2 * n * 16
for label, env in (("native", "0"), ("real planes", "1")):
    os.environ["MXLO_SPARSE_COMPLEX_PLANES"] = env
    opc = lo.LinearOperatorFromMatrix(Mc)
    out = []
    for o in (opc, lo.transpose(opc), lo.adjoint(opc)):
        ms = timeit(lambda: lo.mul(yc, o, xc, 1.0, 0.0), 10)
        out.append(f"{ms * 1e3:7.1f} us {nbc / ms / 1e6 / 8000:5.3f}")
    print(f"complex128 {cases[0][0]} ({label}):  A*x {out[0]}   transpose(A)*x {out[1]}   A'*x {out[2]}", flush=True)
os.environ.pop("MXLO_SPARSE_COMPLEX_PLANES", None)
