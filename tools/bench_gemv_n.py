"""Dense M*v (N mode): the single-launch row-band kernel (tune gemv_n_rows = 1) against the two-launch column-chunk
schedule (0), and the transposed apply beside them. python tools/bench_gemv_n.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
lo = g.load_package()
from linearoperators_jl_amd.device import Timer, get_ctx
dev = torch.device("cuda", 0); ctx = get_ctx(dev); tm = Timer(ctx)


def timeit(fn, reps):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.05:
        for _ in range(10): fn()
        torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        tm.start()
        for _ in range(reps): fn()
        tm.stop()
        best = min(best, tm.elapsed_ms() / reps)
    return best


for dt, es in ((torch.float64, 8), (torch.float32, 4)):
    for m, n in ((2048, 2048), (4096, 4096), (4096, 16384), (8192, 8192), (8192, 32768), (16384, 4096), (16384, 16384), (32768, 8192), (65536, 4096), (262144, 1024)):
        M = torch.rand(n, m, dtype=dt, device=dev).t()
        op = lo.LinearOperatorFromMatrix(M)
        x, y, u = torch.rand(n, dtype=dt, device=dev), torch.empty(m, dtype=dt, device=dev), torch.rand(m, dtype=dt, device=dev)
        z = torch.empty(n, dtype=dt, device=dev)
        nb = float(es) * m * n
        reps = max(20, int(4e-3 / (nb / 6e12)))
        vr = 16 // es
        cols = []
        for rows in (0, 8 * vr, 16 * vr, 32 * vr, 1):
            ctx.tune("gemv_n_rows", rows)
            t = timeit(lambda: lo.mul(y, op, x, 1.0, 0.0), reps)
            cols.append(f"{'chunks+finish' if rows == 0 else ('auto' if rows == 1 else 'RB=%d' % rows)} {t*1e3:7.1f} us ({nb/t/1e6/8000:.3f})")
        ctx.tune("gemv_n_rows", 1)
        tt = timeit(lambda: lo.mul(z, op.T, u, 1.0, 0.0), reps)
        print(f"{str(dt)[6:]} {m:6d} x {n:6d}: " + "  ".join(cols) + f"  | transpose {tt*1e3:7.1f} us ({nb/tt/1e6/8000:.3f})", flush=True)
        del M, op
        torch.cuda.empty_cache()

