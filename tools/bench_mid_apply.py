#!/usr/bin/env python
"""Quasi-Newton applies between the single-launch sizes and the HBM-bound ones (n = 2^19 .. 2^24, fp64): us per mul! and
the rate of the bytes a two-pass apply has to move ((4m + 3) * 8 B per element for the L-BFGS operators with full memory:
2m + 1 vectors read by the dots pass, 2m + 1 read and one written by the combine pass; L-SR1: (2m + 3) * 8)."""
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
gen = torch.Generator(device=dev).manual_seed(5)


def rnd(n):
    return torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 2 - 1


def timeit(fn, reps):
    for _ in range(5):
        fn()
    tm.start()
    for _ in range(reps):
        fn()
    tm.stop()
    return tm.elapsed_ms() / reps * 1e3


persist = os.environ.get("MXLO_QN_PERSIST")
if persist is not None:
    ctx.tune("qn_persist", int(persist))
    print(f"# qn_persist = {persist}")
for key in ("qn_persist_reverse", "qn_persist_prefetch", "qn_persist_lds", "qn_persist_lds_pad", "qn_persist_max_bytes", "qn_persist_min_n"):
    if os.environ.get("MXLO_" + key.upper()) is not None:
        ctx.tune(key, int(os.environ["MXLO_" + key.upper()]))
        print(f"# {key} = {os.environ['MXLO_' + key.upper()]}")
only = os.environ.get("MXLO_MID_ONLY")      # e.g. "inv:5,fwd:10"
if os.environ.get("MXLO_COMBINE_REVERSE") is not None:
    ctx.tune("combine_reverse", int(os.environ["MXLO_COMBINE_REVERSE"]))
    print(f"# combine_reverse = {os.environ['MXLO_COMBINE_REVERSE']}")
for kind, m in (("inv", 5), ("inv", 10), ("inv", 20), ("fwd", 5), ("fwd", 10), ("fwd", 20), ("lsr1", 5), ("lsr1", 20)):
    for e in (19, 20, 21, 22, 23, 24):
        if m * (1 << e) * 8 * 3 > 40e9 or (only and f"{kind}:{m}" not in only.split(",")):
            continue
        n = 1 << e
        op = {"inv": lo.InverseLBFGSOperator, "fwd": lo.LBFGSOperator, "lsr1": lo.LSR1Operator}[kind](torch.float64, n, mem=m, device=dev)
        for _ in range(m + 1):
            s = rnd(n)
            lo.push(op, s, s * (rnd(n) * 0.25 + 1.25) + (0.3 * rnd(n) if kind == "lsr1" else 0))
        x, r = rnd(n), rnd(n)
        us = timeit(lambda: lo.mul(r, op, x, 1.0, 0.0), 300 if e < 22 else 100)
        cols = (2 * m + 3) if kind == "lsr1" else (4 * m + 3)
        gb = cols * 8.0 * n / 1e9
        print(f"{kind:4s} m={m:2d} n=2^{e}: {us:8.1f} us   {gb / us * 1e3:5.2f} TB/s of the two-pass bytes ({gb / us * 1e3 / 8.0:.2f} of peak)", flush=True)
        del op
