"""red_blocks_per_cu (grid of the reduction kernels, workgroups per CU) at HBM sizes: Householder n = 1e8 and the quasi-Newton
applies at n = 5e7. python tools/sweep_red_grid.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
lo = g.load_package()
from linearoperators_jl_amd.device import Timer, get_ctx
dev = torch.device("cuda", 0); ctx = get_ctx(dev); tm = Timer(ctx)


def timeit(fn, reps=10):
    for _ in range(3): fn()
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
H = lo.opHouseholder(h)
for k in (1, 2, 4, 8, 16):
    ctx.tune("red_blocks_per_cu", k)
    ms = timeit(lambda: lo.mul(r, H, v, 1.0, 0.0), 20)
    print(f"opHouseholder n=1e8 red_blocks_per_cu={k:2d}: {ms*1e3:7.1f} us ({40.0*n/ms/1e6/8000:.3f})", flush=True)
ctx.tune("red_blocks_per_cu", 4)
del h, v, r, H
torch.cuda.empty_cache()
n = 50_000_000
for kind, make, m, bpe in (("inv", lo.InverseLBFGSOperator, 10, 344), ("fwd", lo.LBFGSOperator, 20, 664), ("lsr1", lo.LSR1Operator, 10, 184)):
    op = make(torch.float64, n, mem=m, device=dev)
    gen = torch.Generator(device=dev).manual_seed(1)
    for _ in range(m + 1):
        s = torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 2 - 1
        y = s * (torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 0.25 + 1.25) + (0.3 * (torch.rand(n, dtype=torch.float64, device=dev, generator=gen) - 0.5) if kind == "lsr1" else 0)
        lo.push(op, s, y)
    del s, y
    x = torch.rand(n, dtype=torch.float64, device=dev, generator=gen); r = torch.empty_like(x)
    for k in (1, 2, 4, 8, 16):
        ctx.tune("red_blocks_per_cu", k)
        ms = timeit(lambda: lo.mul(r, op, x, 1.0, 0.0))
        print(f"{kind} m={m} n=5e7 red_blocks_per_cu={k:2d}: {ms*1e3:8.1f} us ({bpe*n/ms/1e6/8000:.3f})", flush=True)
    ctx.tune("red_blocks_per_cu", 4)
    del op, x, r
    torch.cuda.empty_cache()
