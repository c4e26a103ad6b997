#!/bin/bash
# Combine-kernel load depth (MXLO_COMBINE_U columns in flight per lane): apply time of the quasi-Newton operators
# at n = 5e7 for libraries built with -DMXLO_COMBINE_U=4/8/16/20 (variants under csrc/variants/).
cd "$(dirname "$0")/.."
for lib in "" linearoperators.jl_amd/csrc/variants/libmxlo_u4.so linearoperators.jl_amd/csrc/variants/libmxlo_u16.so linearoperators.jl_amd/csrc/variants/libmxlo_u20.so; do
  echo "== ${lib:-default (U=8)}"
  MXLO_LIB_PATH=${lib:+$PWD/$lib} python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import torch
import __graft_entry__ as g
lo = g.load_package()
from linearoperators_jl_amd.device import Timer, get_ctx
dev = torch.device("cuda:0"); ctx = get_ctx(dev); tm = Timer(ctx)
n = 50_000_000
gen = torch.Generator(device=dev).manual_seed(1)
rnd = lambda: torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 2 - 1
x, res = rnd(), torch.empty(n, dtype=torch.float64, device=dev)
for kind, m in (("inv", 10), ("fwd", 20), ("lsr1", 10)):
    op = {"inv": lo.InverseLBFGSOperator, "fwd": lo.LBFGSOperator, "lsr1": lo.LSR1Operator}[kind](n, mem=m, device=dev)
    for k in range(m + 1):
        s = rnd(); lo.push(op, s, s * (torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 1.5 + 0.5)); del s
    for _ in range(3): lo.mul(res, op, x, 1.0, 0.0)
    tm.start()
    for _ in range(10): lo.mul(res, op, x, 1.0, 0.0)
    tm.stop()
    print(f"  {kind} m={m}: {tm.elapsed_ms() / 10:7.3f} ms", flush=True)
    del op; torch.cuda.empty_cache()
PY
done
