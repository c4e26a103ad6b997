#!/usr/bin/env python
"""Fixed workload for rocprofv3 counter passes: opHouseholder mul! at n = 2^22 fp64, 4 applies, as ONE launch with two workgroups per CU
(house_fused_per_cu = 2, round 6) or as the dots + update launches (MXLO_HOUSE_FUSED_PER_CU=1)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import __graft_entry__ as g

lo = g.load_package()
dev = torch.device("cuda", 0)
ctx = lo.get_ctx(dev)
ctx.tune("house_fused_per_cu", int(os.environ.get("MXLO_HOUSE_FUSED_PER_CU", "2")))
n = 1 << 22
h = torch.rand(n, dtype=torch.float64, device=dev) - 0.5
h /= torch.linalg.vector_norm(h)
v, res = torch.rand(n, dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.float64, device=dev)
H = lo.opHouseholder(h)
for _ in range(4):
    lo.mul(res, H, v, 1.0, 0.0)
torch.cuda.synchronize()
print("pmc workload house_mid done")
