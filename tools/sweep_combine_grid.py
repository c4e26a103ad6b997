import sys, time, torch
sys.path.insert(0, '/root/repo')
import __graft_entry__ as g
lo = g.load_package()
from linearoperators_jl_amd.device import Timer, get_ctx
dev = torch.device('cuda', 0); ctx = get_ctx(dev); tm = Timer(ctx)
n = 50_000_000
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        tm.start()
        for _ in range(reps): fn()
        tm.stop(); best = min(best, tm.elapsed_ms() / reps)
    return best
for kind, make, m, bpe in (("inv", lo.InverseLBFGSOperator, 10, 344), ("lsr1", lo.LSR1Operator, 10, 184)):
    op = make(torch.float64, n, mem=m, device=dev)
    gen = torch.Generator(device=dev).manual_seed(1)
    for _ in range(m + 1):
        s = torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 2 - 1
        y = s * (torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 0.25 + 1.25) + (0.3 * (torch.rand(n, dtype=torch.float64, device=dev, generator=gen) - 0.5) if kind == "lsr1" else 0)
        lo.push(op, s, y)
    del s, y
    x = torch.rand(n, dtype=torch.float64, device=dev, generator=gen); r = torch.empty_like(x)
    for key, vals in (("combine_blocks_per_cu", (0, 2, 4, 8, 16, 32)), ("combine_reverse", (0, 1))):
        for v in vals:
            ctx.tune(key, v)
            ms = timeit(lambda: lo.mul(r, op, x, 1.0, 0.0))
            print(f"{kind} m={m} n=5e7 {key}={v}: {ms*1e3:8.1f} us  {bpe*n/ms/1e6/8000:.3f}", flush=True)
        ctx.tune(key, 0)
    del op, x, r
    torch.cuda.empty_cache()
