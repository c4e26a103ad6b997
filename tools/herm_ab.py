#!/usr/bin/env python
"""opHermitian A/B inside ONE process: the same operands under several tune settings, interleaved (round 6, second half:
cache policy of the strip loads, single-launch form with the column-block order, finish with one round of loads).
Prints HIP-event time per apply of an eager loop and of a 50-apply hipGraph replay.
    python tools/herm_ab.py [f32] [sizes=4096,8192] [block]"""
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
dt = torch.float32 if "f32" in sys.argv[1:] else torch.float64
es = 4 if dt == torch.float32 else 8
sizes = (2048, 3072, 4096, 5120, 6144, 8192, 16384)
for a in sys.argv[1:]:
    if a.startswith("sizes="):
        sizes = tuple(int(x) for x in a[6:].split(","))
DEFAULT = {"herm_nt": -1, "herm_single": 1, "herm_single_max_n": 0, "herm_order": 1, "herm_strip": 0, "herm_poll_sleep": 4}
configs = [
    ("default", {}),
    ("single<=8192", {"herm_single_max_n": 8192}),
    ("default (again)", {}),
    ("single<=8192 (again)", {"herm_single_max_n": 8192}),
    ("strip 2", {"herm_strip": 2}),
    ("single<=8192 strip 2", {"herm_single_max_n": 8192, "herm_strip": 2}),
]
if os.environ.get("MXLO_HERM_AB_SLEEP"):
    configs = [(f"poll sleep {c}", {"herm_poll_sleep": c}) for c in (32, 8, 2, 1, 64, 32, 8, 2)]
elif os.environ.get("MXLO_HERM_AB_STRIPS"):
    configs = [("default", {})] + [(f"strip {c}", {"herm_strip": c}) for c in (1, 2, 8)]


def apply_cfg(c):
    for k, v in DEFAULT.items():
        ctx.tune(k, c.get(k, v))


def timed(fn, reps):
    for _ in range(5):
        fn()
    best = 1e30
    for _ in range(3):
        tm.start()
        for _ in range(reps):
            fn()
        tm.stop()
        best = min(best, tm.elapsed_ms() / reps)
    return best * 1e3


for nn in sizes:
    M = torch.rand(nn, nn, dtype=dt, device=dev).t()
    d, x, y = (torch.rand(nn, dtype=dt, device=dev) for _ in range(3))
    H = lo.opHermitian(d, M)
    tri = es / 2 * nn * nn
    for name, c in configs:
        apply_cfg(c)
        us = timed(lambda: lo.mul(y, H, x, 1.0, 0.0), 100 if nn <= 8192 else 30)
        cap = lo.CapturedSequence(dev)
        ctx.tune("herm_single", 0)               # (a captured apply takes the two-launch form: size its workspace first)
        lo.mul(y, H, x, 1.0, 0.0)
        apply_cfg(c)
        with cap:
            for _ in range(50):
                lo.mul(y, H, x, 1.0, 0.0)
        ug = timed(lambda: cap.replay(), 4) / 50
        del cap
        print(f"opHermitian {str(dt)[6:]} n={nn:6d} {name:24s}: eager {us:7.1f} us ({tri/us/1e3/8000:5.3f})   graph {ug:7.1f} us ({tri/ug/1e3/8000:5.3f})", flush=True)
    if "block" in sys.argv[1:] and nn in (4096, 16384):
        for k in (2, 4):
            V = torch.rand(k, nn, dtype=dt, device=dev).t()
            R = torch.zeros(k, nn, dtype=dt, device=dev).t()
            for name, c in configs[:3]:
                apply_cfg(c)
                us = timed(lambda: lo.mul(R, H, V, 1.0, 0.0), 50 if nn <= 8192 else 20)
                print(f"opHermitian {str(dt)[6:]} n={nn:6d} block k={k} {name:24s}: {us:7.1f} us ({tri/us/1e3/8000:5.3f})", flush=True)
    apply_cfg({})
    del M, H
