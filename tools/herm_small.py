#!/usr/bin/env python
"""opHermitian at mid sizes: per-apply HIP-event time (for rocprofv3 --kernel-trace runs too)."""
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
if os.environ.get("MXLO_HERM_SINGLE") is not None:
    ctx.tune("herm_single", int(os.environ["MXLO_HERM_SINGLE"]))
    print(f"# herm_single = {os.environ['MXLO_HERM_SINGLE']}")
sizes = tuple(int(x) for x in os.environ.get("MXLO_HERM_SIZES", "1024,2048,4096,8192,16384").split(","))
if os.environ.get("MXLO_HERM_ORDER") is not None:
    ctx.tune("herm_order", int(os.environ["MXLO_HERM_ORDER"]))
    print(f"# herm_order = {os.environ['MXLO_HERM_ORDER']}")
if os.environ.get("MXLO_HERM_LDS_PAD") is not None:
    ctx.tune("herm_lds_pad", int(os.environ["MXLO_HERM_LDS_PAD"]))
    print(f"# herm_lds_pad = {os.environ['MXLO_HERM_LDS_PAD']}")
for nn in sizes:
    M = torch.rand(nn, nn, dtype=dt, device=dev).t()
    d, x, y = (torch.rand(nn, dtype=dt, device=dev) for _ in range(3))
    H = lo.opHermitian(d, M)
    for _ in range(5):
        lo.mul(y, H, x, 1.0, 0.0)
    tm.start()
    for _ in range(50):
        lo.mul(y, H, x, 1.0, 0.0)
    tm.stop()
    ms = tm.elapsed_ms() / 50
    print(f"opHermitian {str(dt)[6:]} n={nn:6d}: {ms*1e3:8.1f} us  {es/2*nn*nn/ms/1e6:7.0f} GB/s  {es/2*nn*nn/ms/1e6/8000:5.3f}", flush=True)
