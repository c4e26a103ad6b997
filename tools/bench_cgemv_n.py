import sys, time, torch
sys.path.insert(0, '/root/repo')
import __graft_entry__ as g
lo = g.load_package()
from linearoperators_jl_amd.device import Timer, get_ctx
dev = torch.device('cuda', 0); ctx = get_ctx(dev); tm = Timer(ctx)
def timeit(fn, reps):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.05:
        for _ in range(10): fn()
        torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        tm.start()
        for _ in range(reps): fn()
        tm.stop(); best = min(best, tm.elapsed_ms() / reps)
    return best
for cdt, rdt, es in ((torch.complex128, torch.float64, 16), (torch.complex64, torch.float32, 8)):
    for m, n in ((4096, 4096), (8192, 8192), (16384, 8192), (16384, 16384)):
        Mc = torch.complex(torch.rand(n, m, dtype=rdt, device=dev), torch.rand(n, m, dtype=rdt, device=dev)).t()
        op = lo.LinearOperatorFromMatrix(Mc)
        x = torch.complex(torch.rand(n, dtype=rdt, device=dev), torch.rand(n, dtype=rdt, device=dev)); y = torch.empty(m, dtype=cdt, device=dev)
        u = torch.complex(torch.rand(m, dtype=rdt, device=dev), torch.rand(m, dtype=rdt, device=dev)); z = torch.empty(n, dtype=cdt, device=dev)
        nb = float(es) * m * n
        reps = max(20, int(4e-3 / (nb / 6e12)))
        t = []
        for rows in (1, 0):
            ctx.tune("gemv_n_rows", rows)
            t.append(timeit(lambda: lo.mul(y, op, x, 1.0, 0.0), reps))
        ctx.tune("gemv_n_rows", 1)
        ta = timeit(lambda: lo.mul(z, op.H, u, 1.0, 0.0), reps)
        print(f"{str(cdt)[6:]} {m:6d} x {n:6d}: M*v row bands {t[0]*1e3:7.1f} us ({nb/t[0]/1e6/8000:.3f})  chunks+finish {t[1]*1e3:7.1f} us ({nb/t[1]/1e6/8000:.3f})  | M'*u {ta*1e3:7.1f} us ({nb/ta/1e6/8000:.3f})", flush=True)
        del Mc, op
        torch.cuda.empty_cache()
