"""rocprofv3 workload: opHermitian / dense GEMV applies at n = 16384 and 4096 (kernel split of the single-pass form)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
lo = g.load_package()
dev = torch.device("cuda:0")
for n in (16384, 4096):
    A = torch.randn(n, n, dtype=torch.float64, device=dev).t().contiguous().t()   # column-major
    d = torch.randn(n, dtype=torch.float64, device=dev)
    v = torch.randn(n, dtype=torch.float64, device=dev)
    res = torch.empty_like(v)
    H = lo.opHermitian(d, A)
    M = lo.LinearOperatorFromMatrix(A)
    for _ in range(10):
        lo.mul(res, H, v, 1.0, 0.0)
        lo.mul(res, M, v, 1.0, 0.0)
        lo.mul(res, M.T, v, 1.0, 0.0)
    torch.cuda.synchronize()
