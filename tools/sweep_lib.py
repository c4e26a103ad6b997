#!/usr/bin/env python
"""GPU-side sweep of libmxlo's runtime knobs (mxlo_ctx_tune) on the BASELINE workloads.
Writes a table to stdout; used to pick defaults (see profiles/)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import __graft_entry__ as g

lo = g.load_package()
from linearoperators_jl_amd import _lib
from linearoperators_jl_amd.device import Timer, dtype_code, get_ctx, ptr

dev = torch.device("cuda", 0)
ctx = get_ctx(dev)
tm = Timer(ctx)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
gen = torch.Generator(device=dev).manual_seed(1)
h = torch.rand(n, dtype=torch.float64, device=dev, generator=gen) - 0.5
h /= torch.linalg.vector_norm(h)
v = torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 2 - 1
res = torch.rand(n, dtype=torch.float64, device=dev, generator=gen)
dot = torch.zeros(1, dtype=torch.float64, device=dev)
f64 = dtype_code(torch.float64)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    tm.start()
    for _ in range(reps):
        fn()
    tm.stop()
    return tm.elapsed_ms() / reps


H, D = lo.opHouseholder(h), lo.opDiagonal(h)
print(f"n={n}")
for nt in (0, 1 << 62):
    ctx.tune("nt_min_bytes", nt)
    for bpc in (0, 8, 16, 32):
        ctx.tune("blocks_per_cu", bpc)
        for rbpc in (2, 4, 8, 16):
            ctx.tune("red_blocks_per_cu", rbpc)
            if bpc != 0 and rbpc != 4:
                continue
            ms_dot = timeit(lambda: _lib.call("mxlo_dot", ctx.handle, f64, ptr(h), ptr(v), n, ptr(dot)))
            row = f"nt={'on ' if nt == 0 else 'off'} blocks_per_cu={bpc:2d} red_blocks_per_cu={rbpc:2d} | dot {16*n/ms_dot/1e6:7.0f} GB/s"
            if rbpc == 4:
                ms_d0 = timeit(lambda: lo.mul(res, D, v, 1.0, 0.0))
                ms_d1 = timeit(lambda: lo.mul(res, D, v, 2.0, -3.0))
                row += f" | diag b0 {24*n/ms_d0/1e6:7.0f} diag b!=0 {32*n/ms_d1/1e6:7.0f}"
                for rev in (0, 1):
                    ctx.tune("house_reverse", rev)
                    ms_u = timeit(lambda: _lib.call("mxlo_householder_apply", ctx.handle, f64, ptr(res), ptr(h), ptr(v), n, 1.0, 0.0, 0, ptr(dot)))
                    ms_h = timeit(lambda: lo.mul(res, H, v, 1.0, 0.0))
                    row += f" | rev={rev}: upd {24*n/ms_u/1e6:7.0f} house {40*n/ms_h/1e6:7.0f} ({ms_h:.4f} ms)"
            print(row, flush=True)
ctx.tune("nt_min_bytes", 256 << 20); ctx.tune("blocks_per_cu", 0); ctx.tune("red_blocks_per_cu", 4); ctx.tune("house_reverse", 1)
del H, D, h, v, res
torch.cuda.empty_cache()

# L-BFGS (config 3: m=10, n=5e7; config 5 shard: m=20, n=5e7)
n = 50_000_000
for kind, m in (("inv", 10), ("fwd", 20), ("lsr1", 10)):
    op = {"inv": lo.InverseLBFGSOperator, "fwd": lo.LBFGSOperator, "lsr1": lo.LSR1Operator}[kind](torch.float64, n, mem=m, device=dev)
    t0 = time.perf_counter()
    for _ in range(m + 2):
        s = torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 2 - 1
        y = (torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 1.5 + 0.5) * s
        lo.push(op, s, y)
        del s, y
    torch.cuda.synchronize()
    tpush = (time.perf_counter() - t0) / (m + 2)
    x = torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 2 - 1
    out = torch.empty_like(x)
    modes = ("twopass", "reforder") if kind == "inv" else ("-",)
    for mode in modes:
        if kind == "inv":
            op.set_mode(mode)
        for nc in (20, 10, 8, 5):
            ctx.tune("dots_max_nc", nc)
            for rbpc in (2, 4, 8):
                ctx.tune("red_blocks_per_cu", rbpc)
                ms = timeit(lambda: lo.mul(out, op, x, 1.0, 0.0), reps=5)
                ncols = (2 * m if kind != "lsr1" else m)
                alg = (2 * ncols + 3) * 8.0 * n
                print(f"{kind} m={m} mode={mode} dots_max_nc={nc:2d} red_bpc={rbpc}: {ms:8.3f} ms/apply  {1e3/ms:7.1f} apply/s  "
                      f"{alg/ms/1e6:7.0f} GB/s (alg {(2*ncols+3)*8} B/elt)  push {tpush*1e3:.1f} ms", flush=True)
                if mode == "reforder":
                    break
            if mode == "reforder":
                break
    ctx.tune("dots_max_nc", 20); ctx.tune("red_blocks_per_cu", 4)
    del op, x, out
    torch.cuda.empty_cache()
