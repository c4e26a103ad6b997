"""kron 1024^2 / 512^2 / 2048^2 f64 under the GEMM tile / wave-layout tune keys."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
lo = g.load_package()
from linearoperators_jl_amd.device import Timer, get_ctx
dev = torch.device("cuda:0")
ctx = get_ctx(dev)
tm = Timer(ctx)
for n in (512, 1024, 2048):
    A = (torch.rand(n, n, dtype=torch.float64, device=dev) - 0.5).t().contiguous().t()
    B = (torch.rand(n, n, dtype=torch.float64, device=dev) - 0.5).t().contiguous().t()
    K = lo.kron(A, B)
    x = torch.rand(n * n, dtype=torch.float64, device=dev)
    r = torch.empty(n * n, dtype=torch.float64, device=dev)
    ref = None
    for tile, waves in ((0, 0), (32, 16), (64, 16), (0, 8), (64, 4)):
        ctx.tune("gemm_tile_m", tile); ctx.tune("gemm_waves", waves)
        for _ in range(5):
            lo.mul(r, K, x, 1.0, 0.0)
        tm.start()
        for _ in range(50):
            lo.mul(r, K, x, 1.0, 0.0)
        tm.stop()
        us = tm.elapsed_ms() / 50 * 1e3
        if ref is None:
            ref = r.clone()
        err = float((r - ref).abs().max())
        print(f"n={n} tile_m={tile:2d} waves={waves:2d}: {us:8.1f} us/apply  {4 * n**3 / us / 1e6:6.1f} TF   maxdiff vs first {err:.1e}", flush=True)
    ctx.tune("gemm_tile_m", 0); ctx.tune("gemm_waves", 0)
