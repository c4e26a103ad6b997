#!/usr/bin/env python
"""Workload for the rocprofv3 --hip-trace view of the allocation / synchronisation contract (tools/contract_trace.sh).

Setup + warm-up, then --iters iterations of the inner loop of a quasi-Newton method on preallocated vectors:
    mul!(res, B, x); mul!(res, Hinv, x); mul!(res, H*D + B, x); diag!(B, d); solve_shifted_system!(x2, B, b, sigma)
and --pushes push!(B, s, y). Running it twice with different --iters / --pushes and subtracting the per-API call
counts isolates what ONE warmed iteration and ONE push! ask of the HIP runtime (process start-up, torch's own
initialisation and the warm-up cancel)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import __graft_entry__ as g

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=100)
ap.add_argument("--pushes", type=int, default=10)
args = ap.parse_args()
lo = g.load_package()
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
n, mem = 1_000_003, 5
T = lambda a: torch.from_numpy(a).to(dev)
B = lo.LBFGSOperator(torch.float64, n, mem=mem, device=dev)
Hi = lo.InverseLBFGSOperator(torch.float64, n, mem=mem, device=dev)
pairs = []
for _ in range(4):
    s = rng.uniform(-1, 1, n)
    pairs.append((T(s), T(s * rng.uniform(0.5, 2.0, n))))
for k in range(mem + 2):
    lo.push(B, *pairs[k % 4])
    lo.push(Hi, *pairs[k % 4])
h = rng.standard_normal(n)
comp = lo.opHouseholder(T(h / np.linalg.norm(h))) * lo.opDiagonal(T(rng.standard_normal(n))) + B
x, b = T(rng.uniform(-1, 1, n)), T(rng.uniform(-1, 1, n))
res, d, x2 = (torch.empty(n, dtype=torch.float64, device=dev) for _ in range(3))


def iteration():
    lo.mul(res, B, x, 1.0, 0.0)
    lo.mul(res, Hi, x, -1.0, 0.0)
    lo.mul(res, comp, x, 2.0, -3.0)
    lo.diag(B, d)
    lo.solve_shifted_system(x2, B, b, 0.25)


for _ in range(3):
    iteration()
torch.cuda.synchronize()
for _ in range(args.iters):
    iteration()
for k in range(args.pushes):
    lo.push(B, *pairs[k % 4])
torch.cuda.synchronize()
print(f"done iters={args.iters} pushes={args.pushes}", flush=True)
