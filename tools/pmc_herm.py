#!/usr/bin/env python
"""Workload of the opHermitian PMC passes: three applies at n = 16384, Float64 then ComplexF64 (triangles of 1.07 / 2.15
GB: well past the 256 MiB Infinity Cache).   rocprofv3 --kernel-trace --output-format csv --pmc <counters> -- python tools/pmc_herm.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import __graft_entry__ as g

lo = g.load_package()
dev = torch.device("cuda", 0)
n = 16384
for cdt in (torch.float64, torch.complex128):
    M = torch.rand(n, n, dtype=torch.float64, device=dev)
    if cdt.is_complex:
        M = torch.complex(M, torch.rand(n, n, dtype=torch.float64, device=dev))
    M = M.t()
    mk = lambda: torch.rand(n, dtype=torch.float64, device=dev).to(cdt)
    d, x, y = torch.rand(n, dtype=torch.float64, device=dev), mk(), mk()
    H = lo.opHermitian(d, M)
    for _ in range(3):
        lo.mul(y, H, x, 1.0, 0.0)
    torch.cuda.synchronize()
    del M, H
