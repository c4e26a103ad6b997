#!/usr/bin/env python
"""Fixed workload for rocprofv3 counter passes over the persistent quasi-Newton apply (qn_apply_persist_kernel): inverse L-BFGS
m = 10 and forward m = 5 at n = 2^20, L-SR1 m = 20 at n = 2^19, 4 applies each, with LDS parking on or off (MXLO_QN_PERSIST_LDS)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import __graft_entry__ as g

lo = g.load_package()
dev = torch.device("cuda", 0)
ctx = lo.get_ctx(dev)
ctx.tune("qn_persist_lds", int(os.environ.get("MXLO_QN_PERSIST_LDS", "1")))
gen = torch.Generator(device=dev).manual_seed(1)
for kind, ctor, n, m in (("inv", lo.InverseLBFGSOperator, 1 << 20, 10), ("fwd", lo.LBFGSOperator, 1 << 20, 5), ("lsr1", lo.LSR1Operator, 1 << 19, 20)):
    op = ctor(torch.float64, n, mem=m, device=dev)
    for _ in range(m + 2):
        s = torch.rand(n, dtype=torch.float64, device=dev, generator=gen) - 0.5
        y = s * (1.0 + 0.5 * torch.rand(n, dtype=torch.float64, device=dev, generator=gen))
        if kind == "lsr1":
            y = y + 0.1 * (torch.rand(n, dtype=torch.float64, device=dev, generator=gen) - 0.5)
        lo.push(op, s, y)
    x, r = torch.rand(n, dtype=torch.float64, device=dev, generator=gen), torch.empty(n, dtype=torch.float64, device=dev)
    for _ in range(4):
        lo.mul(r, op, x, 1.0, 0.0)
    torch.cuda.synchronize()
print("pmc workload persist done")
