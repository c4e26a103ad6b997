import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
lo = g.load_package()
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(1)
n = 50_000_000
kind = sys.argv[1] if len(sys.argv) > 1 else "fwd"
m = int(sys.argv[2]) if len(sys.argv) > 2 else 20
op = {"fwd": lo.LBFGSOperator, "lsr1": lo.LSR1Operator, "inv": lo.InverseLBFGSOperator}[kind](torch.float64, n, mem=m, device=dev)
S = [torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 2 - 1 for _ in range(2)]
Y = [(torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 1.5 + 0.5) * s + (0.3 * (torch.rand(n, dtype=torch.float64, device=dev, generator=gen) - 0.5) if kind == "lsr1" else 0) for s in S]
for i in range(m + 3):
    lo.push(op, S[i % 2], Y[i % 2])
torch.cuda.synchronize()
