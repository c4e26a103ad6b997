import sys, time, torch
sys.path.insert(0, '/root/repo')
import __graft_entry__ as g
lo = g.load_package()
from linearoperators_jl_amd.device import Timer, get_ctx
dev = torch.device('cuda', 0); ctx = get_ctx(dev); tm = Timer(ctx)
def timeit(fn, reps=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        tm.start()
        for _ in range(reps): fn()
        tm.stop(); best = min(best, tm.elapsed_ms() / reps)
    return best
n = 100_000_000
h = torch.rand(n, dtype=torch.float64, device=dev); h /= h.norm()
v = torch.rand(n, dtype=torch.float64, device=dev); r = torch.empty_like(v)
H = lo.opHouseholder(h); D = lo.opDiagonal(h)
for rep in range(2):
    for rev in (1, 0):
        ctx.tune("house_reverse", rev)
        out = []
        for bpc in (0, 1, 2, 4, 8, 16):
            ctx.tune("blocks_per_cu", bpc)
            ms = timeit(lambda: lo.mul(r, H, v, 1.0, 0.0))
            out.append(f"bpc={bpc}: {ms*1e3:6.1f}")
        print(f"householder reverse={rev}  " + "  ".join(out), flush=True)
    ctx.tune("house_reverse", 1)
    out = []
    for bpc in (0, 1, 2, 4, 8, 16):
        ctx.tune("blocks_per_cu", bpc)
        ms = timeit(lambda: lo.mul(r, D, v, 1.0, 0.0))
        out.append(f"bpc={bpc}: {ms*1e3:6.1f}")
    print("opDiagonal            " + "  ".join(out), flush=True)
    ctx.tune("blocks_per_cu", 0)
