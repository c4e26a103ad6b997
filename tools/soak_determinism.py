import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import __graft_entry__ as g
lo = g.load_package()
dev = torch.device("cuda:0")
torch.manual_seed(0)
bad = 0
for n, reps in ((4096, 20000), (65_537, 5000), (1_000_003, 1000), (20_000_001, 100)):
    h = torch.randn(n, dtype=torch.float64, device=dev); h /= h.norm()
    v = torch.randn(n, dtype=torch.float64, device=dev)
    H = lo.opHouseholder(h)
    B = lo.LBFGSOperator(n, mem=7, device=dev); Hi = lo.InverseLBFGSOperator(n, mem=7, device=dev)
    for _ in range(9):
        s = torch.randn(n, dtype=torch.float64, device=dev)
        y = s * (torch.rand(n, dtype=torch.float64, device=dev) + 0.5)
        lo.push(B, s, y); lo.push(Hi, s, y)
    ref = [(op * v).clone() for op in (H, B, Hi)]
    out = torch.empty_like(v)
    for r in range(reps):
        for k, op in enumerate((H, B, Hi)):
            lo.mul(out, op, v)
            if not torch.equal(out, ref[k]):
                bad += 1
    torch.cuda.synchronize()
    print(f"n={n}: {reps} repeats x 3 operators, mismatches so far: {bad}", flush=True)
print("SOAK", "OK" if bad == 0 else "FAILED")
