"""Debug aid: replay one fuzz seed of tests/test_gpu_qn_fuzz.py with the one-pass and the two-kernel push! and print where
their results first differ (every mul!/diag!/solve result of the sequence is logged)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import __graft_entry__ as g
lo = g.load_package()
from linearoperators_jl_amd.device import get_ctx
import test_gpu_qn_fuzz as F
from tolerances import QN_F32, QN_F32_SOLVE
dev = torch.device("cuda", 0)
ctx = get_ctx(dev)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1680
logs = {}
for fused in (1, 0):
    ctx.tune("push_fused", fused)
    log = []
    real_mul, real_solve, real_push = lo.mul, lo.solve_shifted_system, lo.push

    class Proxy:
        def __getattr__(self, k):
            return getattr(lo, k)
        def mul(self, res, op, v, *a):
            out = real_mul(res, op, v, *a); log.append(("mul", res.detach().cpu().numpy().copy())); return out
        def solve_shifted_system(self, x, op, b, sig):
            out = real_solve(x, op, b, sig); log.append(("solve", out.detach().cpu().numpy().copy(), float(sig))); return out
        def push(self, op, s, y, *a):
            out = real_push(op, s, y, *a)
            log.append(("push", np.array([op.data.insert, op.data.scaling_factor]), bool(getattr(op, "_last_push_accepted", True))))
            return out
    try:
        F.run_sequence(Proxy(), dev, seed, torch.float32, 1e9, 1e9)
    except AssertionError as e:
        print("assert", fused, e)
    logs[fused] = log
a, b = logs[1], logs[0]
print("entries", len(a), len(b))
for k, (ea, eb) in enumerate(zip(a, b)):
    d = np.abs(np.asarray(ea[1], np.float64) - np.asarray(eb[1], np.float64)).max() / (np.abs(np.asarray(eb[1], np.float64)).max() + 1e-300)
    flag = " <<<<" if d > 1e-4 else ""
    print(k, ea[0], f"rel diff {d:.3e}", ea[2:] if len(ea) > 2 else "", flag)
