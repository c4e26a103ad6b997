import os, sys
sys.path.insert(0, "/root/repo")
import torch
import __graft_entry__ as g
lo = g.load_package()
from linearoperators_jl_amd.device import Timer, get_ctx
dev = torch.device("cuda", 0); ctx = get_ctx(dev); tm = Timer(ctx)
gen = torch.Generator(device=dev).manual_seed(4)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    tm.start()
    for _ in range(reps): fn()
    tm.stop(); return tm.elapsed_ms()/reps
for bs in (97_657, 97_664, 97_658, 65_536):
    nb = 1024
    dall = torch.rand(nb*bs, dtype=torch.float64, device=dev, generator=gen) + 0.5
    BD = lo.BlockDiagonalOperator(*[lo.opDiagonal(dall[k*bs:(k+1)*bs]) for k in range(nb)])
    x = torch.rand(nb*bs, dtype=torch.float64, device=dev, generator=gen); res = torch.empty_like(x)
    ms = timeit(lambda: lo.mul(res, BD, x, 1.0, 0.0))
    D = lo.opDiagonal(dall)
    ms2 = timeit(lambda: lo.mul(res, D, x, 1.0, 0.0))
    print(f"blocks 1024 x {bs}: blockdiag {24*nb*bs/ms/1e6:.0f} GB/s   same data as ONE opDiagonal {24*nb*bs/ms2/1e6:.0f} GB/s", flush=True)
    del BD, D, dall, x, res
