"""Householder mul! and mxlo_dot with the in-kernel (last-workgroup) finalize on / off, n from 2^12 to 1e8."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
lo = g.load_package()
from linearoperators_jl_amd.device import Timer, get_ctx
dev = torch.device("cuda:0")
ctx = get_ctx(dev)
tm = Timer(ctx)
for n in (1 << 12, 1 << 16, 1 << 20, 10_000_000, 100_000_000):
    h = torch.rand(n, dtype=torch.float64, device=dev) - 0.5
    h /= torch.linalg.vector_norm(h)
    v = torch.rand(n, dtype=torch.float64, device=dev)
    r = torch.empty_like(v)
    H = lo.opHouseholder(h)
    outs = {}
    for fuse in (0, 1, 0, 1):
        ctx.tune("fuse_finalize", fuse)
        reps = 2000 if n <= (1 << 20) else 100
        for _ in range(10):
            lo.mul(r, H, v, 1.0, 0.0)
        tm.start()
        for _ in range(reps):
            lo.mul(r, H, v, 1.0, 0.0)
        tm.stop()
        us = tm.elapsed_ms() / reps * 1e3
        outs[fuse] = r.clone()
        print(f"n={n:>11d} fuse={fuse}: {us:9.2f} us/apply  {40.0 * n / us / 1e3:8.1f} GB/s", flush=True)
    print("   bit-identical fused vs separate finalize:", bool(torch.equal(outs[0], outs[1])), flush=True)
ctx.tune("fuse_finalize", 1)
