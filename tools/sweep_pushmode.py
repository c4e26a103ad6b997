"""Forward L-BFGS at n = 5e7: push! and mul! cost per push mode (gram / compact / reforder at m = 10 only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
lo = g.load_package()
from linearoperators_jl_amd.device import Timer, get_ctx
dev = torch.device("cuda:0")
ctx = get_ctx(dev)
tm = Timer(ctx)
n = 50_000_000
gen = torch.Generator(device=dev).manual_seed(1)
rnd = lambda: torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 2 - 1
x, res = rnd(), torch.empty(n, dtype=torch.float64, device=dev)
for mem in (10, 20):
    pairs = [(rnd(),) for _ in range(4)]
    pairs = [(s[0], s[0] * (torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 1.5 + 0.5)) for s in pairs]
    for mode in ("gram", "compact") + (("reforder",) if mem == 10 else ()):
        B = lo.LBFGSOperator(n, mem=mem, device=dev).set_push_mode(mode)
        for k in range(mem + 1):
            lo.push(B, *pairs[k % 4])
        torch.cuda.synchronize()
        tm.start()
        for k in range(4):
            lo.push(B, *pairs[k])
        tm.stop()
        push_ms = tm.elapsed_ms() / 4
        for _ in range(3):
            lo.mul(res, B, x, 1.0, 0.0)
        tm.start()
        for _ in range(10):
            lo.mul(res, B, x, 1.0, 0.0)
        tm.stop()
        mul_ms = tm.elapsed_ms() / 10
        print(f"m={mem:2d} {mode:9s}: push! {push_ms:7.2f} ms   mul! {mul_ms:6.3f} ms = {(4 * mem + 3) * 8 * n / mul_ms / 1e6:6.0f} GB/s   "
              f"push+mul {push_ms + mul_ms:6.2f} ms", flush=True)
        del B
