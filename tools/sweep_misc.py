#!/usr/bin/env python
"""GPU-side A/B: Householder MALL-tail hybrid (house_mall_tail_bytes) and fast-GEMM wave count."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import __graft_entry__ as g

lo = g.load_package()
from linearoperators_jl_amd.device import Timer, get_ctx

dev = torch.device("cuda", 0)
ctx = get_ctx(dev)
tm = Timer(ctx)
gen = torch.Generator(device=dev).manual_seed(1)


def timeit(fn, reps=30, rounds=3):
    best = []
    for _ in range(rounds):
        for _ in range(3):
            fn()
        tm.start()
        for _ in range(reps):
            fn()
        tm.stop()
        best.append(tm.elapsed_ms() / reps)
    return min(best), sorted(best)[len(best) // 2]


n = 100_000_000
h = torch.rand(n, dtype=torch.float64, device=dev, generator=gen) - 0.5
h /= torch.linalg.vector_norm(h)
v = torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 2 - 1
res = torch.empty_like(v)
H = lo.opHouseholder(h)
for rev in (0, 1):
    ctx.tune("house_reverse", rev)
    mn, med = timeit(lambda: lo.mul(res, H, v, 1.0, 0.0))
    print(f"householder reverse={rev} mall_tail=0: min {mn:.4f} ms  median {med:.4f} ms", flush=True)
ctx.tune("house_reverse", 1)
del H, h, v, res
torch.cuda.empty_cache()

nn = 1024
A = ((torch.rand(nn, nn, dtype=torch.float64, device=dev, generator=gen) * 2 - 1) / 32).t()
B = ((torch.rand(nn, nn, dtype=torch.float64, device=dev, generator=gen) * 2 - 1) / 32).t()
K = lo.kron(A, B)
x = torch.rand(nn * nn, dtype=torch.float64, device=dev, generator=gen) * 2 - 1
out = torch.empty_like(x)
for rnd in range(2):
    for tile in (32, 64, 128, -1):
        ctx.tune("gemm_tile", tile)
        mn, med = timeit(lambda: lo.mul(out, K, x, 1.0, 0.0), reps=50)
        print(f"kron 1024^2 gemm_tile={tile}: min {mn*1e3:.1f} us ({4*nn**3/mn/1e9:.1f} TF)  median {med*1e3:.1f} us", flush=True)
ctx.tune("gemm_tile", 0)
for sz in (256, 512, 2048):
    A = ((torch.rand(sz, sz, dtype=torch.float64, device=dev, generator=gen) * 2 - 1) / 32).t()
    K = lo.kron(A, A)
    x = torch.rand(sz * sz, dtype=torch.float64, device=dev, generator=gen)
    out = torch.empty_like(x)
    for tile in (128, 64, 32, 0):
        ctx.tune("gemm_tile", tile)
        mn, med = timeit(lambda: lo.mul(out, K, x, 1.0, 0.0), reps=20)
        print(f"kron {sz}^2 gemm_tile={tile}: min {mn*1e3:.1f} us ({4*sz**3/mn/1e9:.1f} TF)", flush=True)
