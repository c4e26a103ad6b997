#!/usr/bin/env python
"""push! in the launch-bound regime (m = 5, n = 2^12 .. 2^16): wall clock per accepted push incl. its device-to-host
decision read-back, posted by a kernel into mapped host memory and polled (push_posted = 1) or copied with
hipMemcpyAsync + hipStreamSynchronize (0). (profiles/r04_push_small_finalize_experiment.txt: the same loop with an experimental push pass that
finalized its own partial sums — slower, not kept.)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import __graft_entry__ as g

lo = g.load_package()
from linearoperators_jl_amd.device import get_ctx

dev = torch.device("cuda", 0)
ctx = get_ctx(dev)
gen = torch.Generator(device=dev).manual_seed(3)


def rnd(n):
    return torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 2 - 1


# python tools/bench_push_small.py [kind m log2n]: one configuration only (for rocprofv3 --kernel-trace --stats)
CASES = [(k, m, 1 << e) for k, m in (("fwd", 5), ("inv", 5), ("lsr1", 5), ("fwd", 20), ("inv", 20)) for e in (12, 14, 16)]
if len(sys.argv) == 4:
    CASES = [(sys.argv[1], int(sys.argv[2]), 1 << int(sys.argv[3]))]
for kind, m, n in CASES:
    if True:
        make = {"inv": lo.InverseLBFGSOperator, "fwd": lo.LBFGSOperator, "lsr1": lo.LSR1Operator}[kind]
        S = [rnd(n) for _ in range(16)]
        Y = [s * (rnd(n) * 0.25 + 1.25) + (0.3 * rnd(n) if kind == "lsr1" else 0) for s in S]
        out = []
        for posted in (1, 0):
            ctx.tune("push_posted", posted)
            op = make(torch.float64, n, mem=m, device=dev)
            for i in range(m + 3):
                lo.push(op, S[i % 16], Y[i % 16])
            best = 1e9
            for _ in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(200):
                    lo.push(op, S[i % 16], Y[i % 16])
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / 200 * 1e6)
            out.append(best)
            del op
        ctx.tune("push_posted", 1)
        print(f"push! {kind:4s} m={m:2d} n=2^{n.bit_length()-1:<2d}: posted read-back {out[0]:6.1f} us | hipMemcpyAsync + stream sync {out[1]:6.1f} us", flush=True)
