#!/usr/bin/env python
"""Block apply of a dense LinearOperator(M): mul!(res, op, V) with k columns through mxlo_gemv_block against k GEMVs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import __graft_entry__ as g

lo = g.load_package()
from linearoperators_jl_amd.device import Timer, get_ctx

dev = torch.device("cuda", 0)
tm = Timer(get_ctx(dev))


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    tm.start()
    for _ in range(reps):
        fn()
    tm.stop()
    return tm.elapsed_ms() / reps * 1e3


for key in ("gemvb_t_lds", "gemvb_n_rows"):
    if os.environ.get("MXLO_" + key.upper()) is not None:
        get_ctx(dev).tune(key, int(os.environ["MXLO_" + key.upper()]))
        print(f"# {key} = {os.environ['MXLO_' + key.upper()]}")
for dt, es in ((torch.float64, 8), (torch.float32, 4)):
    if os.environ.get("MXLO_BLOCK_ONLY_DENSE64") and dt != torch.float64:
        continue
    for n in (4096, 16384):
        M = torch.rand(n, n, dtype=dt, device=dev).t()
        op = lo.LinearOperatorFromMatrix(M)
        for k in (2, 4, 8):
            V = torch.rand(k, n, dtype=dt, device=dev).t()
            R = torch.empty(k, n, dtype=dt, device=dev).t()
            for name, o in (("M*V", op), ("M'*V", op.T)):
                tb = timeit(lambda: lo.mul(R, o, V))
                tc = timeit(lambda: [lo.mul(R[:, j], o, V[:, j]) for j in range(k)])
                print(f"{str(dt)[6:]} n={n:6d} k={k} {name:5s}: block {tb:8.1f} us = {es*n*n/tb/1e3:6.0f} GB/s of M ({es*n*n/tb/1e3/8000:5.3f}),"
                      f" {k} GEMVs {tc:8.1f} us  -> x{tc/tb:4.2f}", flush=True)
        del M, op

if os.environ.get("MXLO_BLOCK_ONLY_DENSE64"):
    sys.exit(0)
# opHermitian on a block of k vectors: mxlo_hermitian_mul_block (triangle read once per 4 columns) against k single applies
for dt, es in ((torch.float64, 8), (torch.float32, 4)):
    for n in (4096, 16384):
        M = torch.rand(n, n, dtype=dt, device=dev).t()
        H = lo.opHermitian(torch.rand(n, dtype=dt, device=dev), M)
        for k in (2, 4, 8):
            V = torch.rand(k, n, dtype=dt, device=dev).t()
            R = torch.empty(k, n, dtype=dt, device=dev).t()
            tb = timeit(lambda: lo.mul(R, H, V))
            tc = timeit(lambda: [lo.mul(R[:, j], H, V[:, j]) for j in range(k)])
            tri = es / 2 * n * n
            passes = (k + 3) // 4 if k >= 4 else 1                  # chunks of up to 4 columns: one pass over the triangle each
            print(f"{str(dt)[6:]} n={n:6d} k={k} opHermitian: block {tb:8.1f} us ({passes} pass(es) over the triangle: {tri*passes/tb/1e3:6.0f} GB/s),"
                  f" {k} single applies {tc:8.1f} us  -> x{tc/tb:4.2f}", flush=True)
        del M, H
