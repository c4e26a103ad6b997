#!/usr/bin/env python
"""Randomised check of opHermitian (Float64 / Float32 / ComplexF64 / ComplexF32) against dense NumPy: random order n
(biased to the row-group boundaries 128 / 256 and to the strip regimes), leading dimension, row offset of the view
(16-byte aligned or not), real or complex diagonal, α / β incl. 0, and NaN planted at and above the diagonal of A (the
reference reads tril(A, -1): nothing up there may reach the result).   python tools/fuzz_hermitian.py [seconds] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import __graft_entry__ as g

lo = g.load_package()
dev = torch.device("cuda", 0)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
DT = [(np.float64, torch.float64, 1e-12), (np.float32, torch.float32, 3e-5), (np.complex128, torch.complex128, 1e-12),
      (np.complex64, torch.complex64, 3e-5)]


def rnd(shape, dt):
    a = rng.uniform(-1, 1, shape)
    if np.issubdtype(dt, np.complexfloating):
        a = a + 1j * rng.uniform(-1, 1, shape)
    return a.astype(dt)


t0, cases, worst = time.time(), 0, 0.0
while time.time() - t0 < budget:
    npd, tdt, tol = DT[int(rng.integers(4))]
    kind = int(rng.integers(6))
    if kind == 0:
        n = int(rng.integers(1, 40))
    elif kind == 1:
        n = int(rng.choice([128, 256, 512, 1024])) * int(rng.integers(1, 4)) + int(rng.integers(-2, 3))
    elif kind == 2:
        n = int(rng.integers(40, 1500))
    elif kind == 3:
        n = int(rng.integers(1500, 3200))
    elif kind == 4:
        n = int(rng.choice([3001, 3072, 4224, 4352, 4500]))          # 2-tile / full-width strips of the complex kernel
    else:
        n = int(rng.choice([2048, 2304, 2560, 4096]))
    n = max(n, 1)
    off, pad = int(rng.integers(0, 3)), int(rng.integers(0, 4))
    big = rnd((n + off + pad, n), npd)                                 # numpy row-major (rows, cols)
    bigd = torch.from_numpy(big.T.copy()).to(dev).t()                  # device column-major, lda = n + off + pad
    A, Ad = big[off:off + n, :], bigd[off:off + n, :]
    if rng.integers(2):                                                # poison the diagonal and everything above it
        iu = np.triu_indices(n)
        A = A.copy()
        A[iu] = np.nan
        Ad = Ad.clone() if False else Ad
        bigd_view = bigd[off:off + n, :]
        mask = torch.from_numpy(np.triu(np.ones((n, n), dtype=bool))).to(dev)
        bigd_view[mask] = float("nan")
    rdt = np.float64 if npd in (np.float64, np.complex128) else np.float32
    d = rnd(n, npd) if rng.integers(2) else rnd(n, rdt)
    if not np.issubdtype(npd, np.complexfloating):
        d = d.astype(npd)
    v, r0 = rnd(n, npd), rnd(n, npd)
    cplx = np.issubdtype(npd, np.complexfloating)
    a = [1.0, 2.0, -0.5, (1.5 - 0.5j) if cplx else 1.5][int(rng.integers(4))]
    b = [0.0, 0.0, -3.0, (0.25 + 2j) if cplx else 0.25][int(rng.integers(4))]
    H = lo.opHermitian(torch.from_numpy(d).to(dev), Ad)
    res = torch.from_numpy(r0.copy()).to(dev)
    lo.mul(res, H, torch.from_numpy(v).to(dev), a, b)
    L = np.tril(np.nan_to_num(A.astype(np.complex128), nan=0.0), -1)
    want = a * (d.astype(np.complex128) * v + L @ v.astype(np.complex128) + L.conj().T @ v.astype(np.complex128)) + (b * r0.astype(np.complex128) if b != 0 else 0)
    got = res.cpu().numpy().astype(np.complex128)
    err = np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-300)
    assert np.isfinite(got).all() and err <= tol * 4, (npd.__name__, n, off, pad, d.dtype, a, b, err)
    worst = max(worst, err / tol)
    cases += 1
print(f"{cases} random opHermitian cases in {time.time() - t0:.0f} s: all within tolerance (worst {worst:.2f} x tol), NaN above the diagonal never propagated", flush=True)
