#!/usr/bin/env python
"""Where the microseconds of a small-n apply go: kernel vs ctypes vs the Python host mirror (the mirror stands in for
Julia's `ccall`, which has no marshalling cost). Wall clock per call over back-to-back calls + HIP-event time."""
import cProfile
import ctypes as C
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import __graft_entry__ as g

lo = g.load_package()
from linearoperators_jl_amd import _lib
from linearoperators_jl_amd.device import Timer, get_ctx, ptr

dev = torch.device("cuda", 0)
ctx = get_ctx(dev)
tm = Timer(ctx)


def wall(fn, reps=5000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def dev_time(fn, reps=2000):
    for _ in range(50):
        fn()
    tm.start()
    for _ in range(reps):
        fn()
    tm.stop()
    return tm.elapsed_ms() / reps * 1e3


for n in (1 << 12, 1 << 16, 1 << 20):
    h = torch.rand(n, dtype=torch.float64, device=dev)
    h /= torch.linalg.vector_norm(h)
    v, res = torch.rand(n, dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.float64, device=dev)
    H, D = lo.opHouseholder(h), lo.opDiagonal(h)
    L = _lib.lib()
    fh = L.mxlo_householder_mul
    args = (ctx.handle, 0, C.c_void_p(res.data_ptr()), C.c_void_p(h.data_ptr()), C.c_void_p(v.data_ptr()), C.c_int64(n),
            C.c_double(1.0), C.c_double(0.0), 0)
    fd = L.mxlo_diag_mul
    argsd = (ctx.handle, 0, C.c_void_p(res.data_ptr()), C.c_void_p(h.data_ptr()), C.c_void_p(v.data_ptr()), C.c_int64(n),
             C.c_int64(n), C.c_double(1.0), C.c_double(0.0), 0)
    for name, full, direct in (("opHouseholder", lambda: lo.mul(res, H, v, 1.0, 0.0), lambda: fh(*args)),
                               ("opDiagonal", lambda: lo.mul(res, D, v, 1.0, 0.0), lambda: fd(*argsd))):
        for tag, fused in (("", 1),) if name == "opDiagonal" else (("single-launch", 1), ("two-launch", 0)):
            ctx.tune("house_fused", fused)
            print(f"n=2^{n.bit_length()-1:<2d} {name:14s} {tag:14s} host mirror lo.mul: wall {wall(full):6.2f} us, HIP events {dev_time(full):6.2f} us | "
                  f"pre-bound ctypes call of the C ABI: wall {wall(direct):6.2f} us, HIP events {dev_time(direct):6.2f} us", flush=True)
        ctx.tune("house_fused", 1)
    g1 = lo.capture_mul(res, H, v, 1.0, 0.0)
    print(f"n=2^{n.bit_length()-1:<2d} opHouseholder graph replay: wall {wall(lambda: g1.replay(sync_streams=False), 3000):6.2f} us", flush=True)

n = 1 << 16
h = torch.rand(n, dtype=torch.float64, device=dev)
v, res = torch.rand(n, dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.float64, device=dev)
H = lo.opHouseholder(h)
pr = cProfile.Profile()
pr.enable()
for _ in range(20000):
    lo.mul(res, H, v, 1.0, 0.0)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
