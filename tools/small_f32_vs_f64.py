import time, ctypes as C, torch, sys
sys.path.insert(0, '/root/repo')
import __graft_entry__ as g
lo = g.load_package()
from linearoperators_jl_amd import _lib
from linearoperators_jl_amd.device import Timer, get_ctx
dev = torch.device('cuda', 0); ctx = get_ctx(dev); tm = Timer(ctx)
def counters():
    a = (C.c_int64 * 12)(); _lib.call("mxlo_debug_counters", a); return list(a)
for dt in (torch.float64, torch.float32):
    for n in (1 << 16, 1 << 20):
        d = torch.rand(n, dtype=dt, device=dev) + 0.5; v = torch.rand(n, dtype=dt, device=dev); r = torch.empty_like(v)
        h = torch.rand(n, dtype=dt, device=dev); h /= h.norm()
        for name, op in (("diag", lo.opDiagonal(d)), ("hh", lo.opHouseholder(h))):
            for _ in range(50): lo.mul(r, op, v, 1.0, 0.0)
            torch.cuda.synchronize()
            c0 = counters(); lo.mul(r, op, v, 1.0, 0.0); c1 = counters()
            t = time.perf_counter()
            for _ in range(2000): lo.mul(r, op, v, 1.0, 0.0)
            host = (time.perf_counter() - t) / 2000 * 1e6
            torch.cuda.synchronize()
            tm.start()
            for _ in range(2000): lo.mul(r, op, v, 1.0, 0.0)
            tm.stop(); eager = tm.elapsed_ms() / 2000 * 1e3
            gq = lo.CapturedSequence(dev)
            with gq:
                for _ in range(500): lo.mul(r, op, v, 1.0, 0.0)
            for _ in range(3): gq.replay()
            torch.cuda.synchronize()
            tm.start(); gq.replay(); tm.stop(); rep = tm.elapsed_ms() / 500 * 1e3
            print(f"{name} {dt} n={n}: launches/call {c1[10]-c0[10]} memset {c1[9]-c0[9]}  host issue {host:.2f} us  eager {eager:.2f} us  replay {rep:.2f} us", flush=True)
