"""Reference point for the kron GEMMs: the vendor library's f64 / f32 GEMM (through torch.mm) at the same shapes,
timed with HIP events next to libmxlo's kron GEMM kernel (gemm_glds_kernel, through mxlo_kron_mul = two GEMMs; tools/tune_gemm.hip is the same-box comparison used in round 2)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
lo = g.load_package()
dev = torch.device("cuda:0")


def time_it(f, it=50):
    for _ in range(5):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(it):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


for dt in (torch.float64, torch.float32):
    for n in (512, 1024, 2048):
        A = torch.rand(n, n, dtype=dt, device=dev) - 0.5
        B = torch.rand(n, n, dtype=dt, device=dev) - 0.5
        C = torch.empty(n, n, dtype=dt, device=dev)
        us = time_it(lambda: torch.mm(A, B, out=C))
        K = lo.kron(A.t().contiguous().t(), B.t().contiguous().t())
        x = torch.rand(n * n, dtype=dt, device=dev)
        r = torch.empty(n * n, dtype=dt, device=dev)
        usk = time_it(lambda: lo.mul(r, K, x, 1.0, 0.0))
        print(f"{dt} n={n}: vendor GEMM {us:8.1f} us = {2 * n**3 / us / 1e6:6.1f} TF | kron (2 GEMMs) {usk:8.1f} us = "
              f"{4 * n**3 / usk / 1e6:6.1f} TF  ({usk / 2:7.1f} us per GEMM)")
