#!/usr/bin/env python
"""What does the kernel after the opHermitian pass cost? apply; small unrelated kernel; repeated (for rocprofv3 traces)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import __graft_entry__ as g

lo = g.load_package()
dev = torch.device("cuda", 0)
nn = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
M = torch.rand(nn, nn, dtype=torch.float64, device=dev).t()
d, x, y, z = (torch.rand(nn, dtype=torch.float64, device=dev) for _ in range(4))
H = lo.opHermitian(d, M)
D = lo.opDiagonal(d)
for _ in range(30):
    lo.mul(y, H, x, 1.0, 0.0)
    lo.mul(z, D, x, 1.0, 0.0)
    lo.mul(z, D, y, 1.0, 0.0)
torch.cuda.synchronize()
