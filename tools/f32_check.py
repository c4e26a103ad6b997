"""fp32 instantiations at HBM-bound sizes: are they as close to the roofline as their fp64 twins?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
lo = g.load_package()
from linearoperators_jl_amd.device import Timer, get_ctx
dev = torch.device("cuda:0"); ctx = get_ctx(dev); tm = Timer(ctx)
gen = torch.Generator(device=dev).manual_seed(1)


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    tm.start()
    for _ in range(reps):
        fn()
    tm.stop()
    return tm.elapsed_ms() / reps


for dt, es in ((torch.float64, 8), (torch.float32, 4)):
    n = 50_000_000 * (8 // es)                       # same bytes per vector for both dtypes
    rnd = lambda k=n: torch.rand(k, dtype=dt, device=dev, generator=gen) * 2 - 1
    x, res = rnd(), torch.empty(n, dtype=dt, device=dev)
    for kind, m in (("inv", 10), ("fwd", 10), ("lsr1", 10)):
        op = {"inv": lo.InverseLBFGSOperator, "fwd": lo.LBFGSOperator, "lsr1": lo.LSR1Operator}[kind](dt, n, mem=m, device=dev)
        for k in range(m + 1):
            s = rnd(); lo.push(op, s, s * (torch.rand(n, dtype=dt, device=dev, generator=gen) * 1.5 + 0.5)); del s
        ncol = 2 * m if kind != "lsr1" else m
        ms = timeit(lambda: lo.mul(res, op, x, 1.0, 0.0))
        print(f"{str(dt):14s} {kind:4s} m={m} n={n:.0e}: {ms:7.3f} ms  {(2 * ncol + 3) * es * n / ms / 1e6:6.0f} GB/s", flush=True)
        del op; torch.cuda.empty_cache()
    del x, res
    nh = 16384
    A = torch.rand(nh, nh, dtype=dt, device=dev, generator=gen).t().contiguous().t()
    d, v = rnd(nh), rnd(nh)
    r = torch.empty(nh, dtype=dt, device=dev)
    H, M = lo.opHermitian(d, A), lo.LinearOperatorFromMatrix(A)
    for name, op, byts in (("opHermitian", H, es * nh * nh / 2), ("gemv N", M, es * nh * nh), ("gemv T", M.T, es * nh * nh)):
        ms = timeit(lambda: lo.mul(r, op, v, 1.0, 0.0), 20)
        print(f"{str(dt):14s} {name:12s} n={nh}: {ms * 1e3:8.1f} us  {byts / ms / 1e6:6.0f} GB/s", flush=True)
    del A, H, M
    torch.cuda.empty_cache()
