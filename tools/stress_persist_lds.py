#!/usr/bin/env python
"""Soak test of the persistent quasi-Newton apply with LDS parking (csrc/qn.hip: qn_apply_persist_kernel, tune key qn_persist_lds):
back-to-back applies over changing operators (inverse / forward L-BFGS, L-SR1; f64 / f32; n = 2^19 ... 2^21 incl. ragged lengths;
mem 2 ... 20; partially filled memories), interleaved with pushes, long streaming launches, Hermitian single launches and graph
replays, every result compared BIT FOR BIT with the same apply under qn_persist_lds = 0 (computed once per operator state) —
a parked tile read back wrongly, or LDS left over from another kernel, would show as a wrong element.
Usage: python tools/stress_persist_lds.py [seconds] [seed]"""
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
ctx.tune("qn_persist_min_bytes", 0)
big = torch.rand(30_000_000, dtype=torch.float64, device=dev)
bigr = torch.empty_like(big)
D = lo.opDiagonal(big)
Mh = torch.rand(2048, 2048, dtype=torch.float64, device=dev).t()
Hh = lo.opHermitian(torch.rand(2048, dtype=torch.float64, device=dev), Mh)
xh, yh = (torch.rand(2048, dtype=torch.float64, device=dev) for _ in range(2))
kinds = {"inv": lo.InverseLBFGSOperator, "fwd": lo.LBFGSOperator, "lsr1": lo.LSR1Operator}
shapes = [(1 << 19, 10), ((1 << 19) + 5, 4), (700_001, 7), (1 << 20, 5), ((1 << 20) + 3, 10), (1 << 20, 20), (1_500_003, 3), ((1 << 21) + 2, 5), (1 << 21, 2)]


def rnd(n, dt):
    return (torch.rand(n, dtype=dt, device=dev) - 0.5)


def push_pair(op, n, dt, kind):
    s = rnd(n, dt)
    y = s * (1.0 + 0.5 * torch.rand(n, dtype=dt, device=dev))
    if kind == "lsr1":
        y = y + 0.1 * rnd(n, dt)
    lo.push(op, s, y)


ops = []
for dt in (torch.float64, torch.float32):
    for kind in kinds:
        for n, mem in shapes:
            if rng.random() < 0.45:
                continue
            op = kinds[kind](dt, n, mem=mem, device=dev)
            for _ in range(int(rng.integers(1, mem + 3))):          # partially filled or wrapped
                push_pair(op, n, dt, kind)
            ops.append({"op": op, "kind": kind, "n": n, "dt": dt, "x": rnd(n, dt), "r0": rnd(n, dt), "ref": None, "out": torch.empty(n, dtype=dt, device=dev)})
print(f"# {len(ops)} operators", flush=True)


def reference(o):
    ctx.tune("qn_persist_lds", 0)
    ref = {}
    for a, b in ((1.0, 0.0), (0.7, -1.3)):
        r = o["r0"].clone()
        lo.mul(r, o["op"], o["x"], a, b)
        ref[(a, b)] = r
    ctx.tune("qn_persist_lds", 1)
    o["ref"] = ref


for o in ops:
    reference(o)
torch.cuda.synchronize()
gop = lo.LBFGSOperator(torch.float64, (1 << 20) + 7, mem=6, device=dev)       # its own operator: a captured apply must not see pushes
for _ in range(8):
    push_pair(gop, (1 << 20) + 7, torch.float64, "fwd")
gx, gout = rnd((1 << 20) + 7, torch.float64), torch.empty((1 << 20) + 7, dtype=torch.float64, device=dev)
ctx.tune("qn_persist_lds", 0)
lo.mul(gout, gop, gx, 1.0, 0.0)
gref = gout.clone()
ctx.tune("qn_persist_lds", 1)
cap = lo.capture_mul(gout, gop, gx, 1.0, 0.0)
t0, n_apply, n_check, n_push = time.perf_counter(), 0, 0, 0
while time.perf_counter() - t0 < budget:
    o = ops[int(rng.integers(len(ops)))]
    r = rng.random()
    if r < 0.08:                                                      # the operator changes: new reference
        push_pair(o["op"], o["n"], o["dt"], o["kind"])
        reference(o)
        n_push += 1
    for _ in range(int(rng.integers(1, 25))):
        lo.mul(o["out"], o["op"], o["x"], 1.0, 0.0)
        n_apply += 1
    r = rng.random()
    if r < 0.15:
        lo.mul(bigr, D, big, 1.0, 0.0)                                # a long streaming launch in between
    elif r < 0.3:
        lo.mul(yh, Hh, xh, 1.0, 0.0)                                  # another single-launch form (its own LDS use)
    elif r < 0.4:
        gout.fill_(float("nan"))
        cap.replay()
        torch.cuda.synchronize()
        assert torch.equal(gout, gref), f"MISMATCH in the graph replay after {n_apply} applies"
        n_check += 1
    for (a, b), want in o["ref"].items():
        o["out"].copy_(o["r0"])
        lo.mul(o["out"], o["op"], o["x"], a, b)
        torch.cuda.synchronize()
        assert torch.equal(o["out"], want), f"MISMATCH after {n_apply} applies: {o['kind']} {o['dt']} n={o['n']} ({a}, {b})"
        n_apply += 1
        n_check += 1
ctx.sync()
ctx.tune("qn_persist_min_bytes", 32 << 20)
print(f"stress_persist_lds: {n_apply} applies with LDS parking over {len(ops)} operators ({n_push} pushes in between), {n_check} bit-exact "
      f"checks against the plain persistent form, {time.perf_counter() - t0:.0f} s: OK")
