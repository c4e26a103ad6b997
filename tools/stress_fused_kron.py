#!/usr/bin/env python
"""Soak test of the one-launch kron (gemm_glds.h: kron_fused_kernel): tens of thousands of back-to-back applies over changing shapes
(square / rectangular, on and off the tile grid, f64 / f32, prod and tprod), interleaved with other kernels, graph replays and long
streaming launches, every result compared BIT FOR BIT with the two-launch schedule's (computed once per operator) — a lost or early
counter update would show as a wrong tile. Usage: python tools/stress_fused_kron.py [seconds] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import __graft_entry__ as g

lo = g.load_package()
dev = torch.device("cuda", 0)
ctx = lo.get_ctx(dev)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
big = torch.rand(50_000_000, dtype=torch.float64, device=dev)
bigr = torch.empty_like(big)
D = lo.opDiagonal(big)
shapes = [((64, 64), (64, 64)), ((128, 128), (128, 128)), ((256, 256), (256, 256)), ((200, 136), (312, 248)), ((512, 512), (512, 512)),
          ((520, 384), (384, 520)), ((768, 512), (512, 640)), ((1024, 1024), (1024, 1024)), ((1000, 1016), (1008, 992)), ((96, 1024), (1024, 96))]
ops = []
for dt in (torch.float64, torch.float32):
    for (am, an), (bp, bq) in shapes:
        A = (torch.rand(an, am, dtype=dt, device=dev) - 0.5).t()
        B = (torch.rand(bq, bp, dtype=dt, device=dev) - 0.5).t()
        K = lo.kron(A, B)
        x, xt = torch.rand(an * bq, dtype=dt, device=dev) - 0.5, torch.rand(am * bp, dtype=dt, device=dev) - 0.5
        ctx.tune("kron_fuse", 0)
        ref = torch.empty(am * bp, dtype=dt, device=dev)
        lo.mul(ref, K, x, 1.0, 0.0)
        reft = torch.empty(an * bq, dtype=dt, device=dev)
        lo.mul(reft, K.T, xt, 1.0, 0.0)
        ctx.tune("kron_fuse", 1)
        ops.append((K, x, xt, ref, reft, torch.empty_like(ref), torch.empty_like(reft), f"{str(dt)[6:]} {am}x{an} (x) {bp}x{bq}"))
torch.cuda.synchronize()
graph = lo.capture_mul(ops[3][5], ops[3][0], ops[3][1], 1.0, 0.0)      # a captured fused apply, replayed in between
t0, n_apply, n_check = time.perf_counter(), 0, 0
while time.perf_counter() - t0 < budget:
    k = int(rng.integers(len(ops)))
    K, x, xt, ref, reft, out, outt, name = ops[k]
    reps = int(rng.integers(1, 40))
    for _ in range(reps):
        if rng.random() < 0.5:
            lo.mul(out, K, x, 1.0, 0.0)
        else:
            lo.mul(outt, K.T, xt, 1.0, 0.0)
        n_apply += 1
    r = rng.random()
    if r < 0.15:
        lo.mul(bigr, D, big, 1.0, 0.0)                     # a long streaming launch between fused ones
    elif r < 0.3:
        graph.replay()
    out.fill_(float("nan"))
    outt.fill_(float("nan"))
    lo.mul(out, K, x, 1.0, 0.0)
    lo.mul(outt, K.T, xt, 1.0, 0.0)
    torch.cuda.synchronize()
    assert torch.equal(out, ref) and torch.equal(outt, reft), f"MISMATCH after {n_apply} applies: {name}"
    n_apply += 2
    n_check += 1
ctx.sync()
print(f"stress_fused_kron: {n_apply} fused applies over {len(ops)} operators, {n_check} bit-exact checks against the two-launch results, "
      f"{time.perf_counter() - t0:.0f} s: OK")
