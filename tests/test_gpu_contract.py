"""-m gpu: the allocation / synchronisation contract, mirroring test/test_lbfgs.jl:180-218
(`@allocated mul!(...) == 0` after a warm-up) at the level that matters on a GPU: what a call asks of the HIP runtime.

A WARMED `mul!`, `diag!`, `solve_shifted_system!` (and every leaf apply) must issue kernel launches and nothing else —
no hipMalloc/hipFree, no copy in any direction, no stream/device/event synchronisation, no memset. `push!` must issue
exactly ONE device-to-host transfer (its few doubles of replicated control state — since round 4 posted into mapped
pinned host memory by a one-wave kernel and polled by the host, `push_posted`; counted as one transfer + one wait, which
is what it is; with `push_posted` = 0 a hipMemcpyAsync + hipStreamSynchronize) with the one wait that transfer
needs, and no allocation. The library counts every allocating / copying / blocking runtime call it makes
(`mxlo_debug_counters`, include/mxlo.h); the rocprofv3 `--hip-trace` view of the same workload is committed under
profiles/ (tools/contract_trace.py)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

NAMES = ("malloc", "free", "h2d", "d2h", "d2d", "d2h_bytes", "stream_sync", "device_sync", "event_sync", "memset_async",
         "launch", "blocking_copy")


def snap(lo):
    a = (C.c_int64 * 12)()
    lo._lib.call("mxlo_debug_counters", a)
    return dict(zip(NAMES, list(a)))


def delta(lo, fn, reps=1):
    import gc
    gc.collect()                 # operators of earlier tests that only the cycle collector can free (closures referring to
    gc.collect()                 # their operator) must not be destroyed — hipFree, stream sync — inside the measured window
    torch.cuda.synchronize()
    a = snap(lo)
    for _ in range(reps):
        fn()
    b = snap(lo)
    torch.cuda.synchronize()
    return {k: b[k] - a[k] for k in NAMES}


def only_launches(d, reps=1, what=""):
    quiet = {k: v for k, v in d.items() if k != "launch" and v}
    assert not quiet, f"{what}: a warmed apply must only launch kernels, got {quiet}"
    assert d["launch"] >= reps, (what, d)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_warmed_applies_only_launch_kernels(lo, dev, dtype):
    rng = np.random.default_rng(0)
    n, mem = 100_003, 5
    T = lambda a: torch.from_numpy(np.asarray(a)).to(dtype).to(dev)
    v, res = T(rng.uniform(-1, 1, n)), T(rng.uniform(-1, 1, n))
    h = rng.standard_normal(n)
    ops = {
        "opDiagonal": lo.opDiagonal(T(rng.standard_normal(n))),
        "opHouseholder (single launch)": lo.opHouseholder(T(h / np.linalg.norm(h))),
        "opEye": lo.opEye(dtype, n, S=lo.Storage(dtype, dev)),
        "opOnes": lo.opOnes(dtype, n, n, S=lo.Storage(dtype, dev)),
    }
    qn = {"InverseLBFGS": lo.InverseLBFGSOperator(dtype, n, mem=mem, device=dev),
          "LBFGS": lo.LBFGSOperator(dtype, n, mem=mem, device=dev),
          "LSR1": lo.LSR1Operator(dtype, n, mem=mem, device=dev)}
    for name, op in qn.items():
        for k in range(mem + 2):
            s = rng.uniform(-1, 1, n)
            lo.push(op, T(s), T(s * rng.uniform(0.5, 2.0, n) + (0.3 * rng.standard_normal(n) if name == "LSR1" else 0)))
        ops[name] = op
    ops["H*D + B (compose + sum)"] = ops["opHouseholder (single launch)"] * ops["opDiagonal"] + ops["LBFGS"]
    ops["ShiftedOperator(B, 0.5)"] = lo.ShiftedOperator(ops["LBFGS"], 0.5)
    for name, op in ops.items():
        for _ in range(3):                                        # warm-up: temporaries of compose/sum, workspaces
            lo.mul(res, op, v, 2.0, -3.0)
            lo.mul(res, op, v, 1.0, 0.0)
        only_launches(delta(lo, lambda: lo.mul(res, op, v, 2.0, -3.0), 4), 4, f"mul! {name} (beta != 0)")
        only_launches(delta(lo, lambda: lo.mul(res, op, v, 1.0, 0.0), 4), 4, f"mul! {name} (beta == 0)")
        if name in ("InverseLBFGS", "LBFGS", "LSR1", "opDiagonal"):
            only_launches(delta(lo, lambda: lo.mul(res, op.T, v, 1.0, 0.0), 2), 2, f"mul! transpose({name})")
    # the two-launch Householder path (what sharded runs use) obeys the same contract
    ctx = lo.get_ctx(dev)
    ctx.tune("house_fused", 0)
    try:
        H = ops["opHouseholder (single launch)"]
        lo.mul(res, H, v, 1.0, 0.0)
        only_launches(delta(lo, lambda: lo.mul(res, H, v, 1.0, 0.0), 3), 3, "mul! opHouseholder (two launches)")
    finally:
        ctx.tune("house_fused", 1)
    # diag! and solve_shifted_system! on warmed operators
    B = qn["LBFGS"]
    dvec = torch.empty(n, dtype=dtype, device=dev)
    lo.diag(B, dvec)
    lo.diag(qn["LSR1"], dvec)
    only_launches(delta(lo, lambda: lo.diag(B, dvec), 2), 2, "diag! LBFGS")
    only_launches(delta(lo, lambda: lo.diag(qn["LSR1"], dvec), 2), 2, "diag! LSR1")
    x = torch.empty(n, dtype=dtype, device=dev)
    lo.solve_shifted_system(x, B, v, 0.25)
    only_launches(delta(lo, lambda: lo.solve_shifted_system(x, B, v, 0.25), 3), 3, "solve_shifted_system! (G cached)")
    lo.solve_shifted_system(x, B, v, 0.75)                               # a new sigma rebuilds G on the device
    only_launches(delta(lo, lambda: lo.solve_shifted_system(x, B, v, 1.25), 1), 1, "solve_shifted_system! (new sigma)")


@pytest.mark.parametrize("kind", ["InverseLBFGS", "LBFGS", "LSR1"])
def test_push_is_one_small_d2h_and_no_allocation(lo, dev, kind):
    rng = np.random.default_rng(1)
    n, mem = 50_001, 4
    make = {"InverseLBFGS": lo.InverseLBFGSOperator, "LBFGS": lo.LBFGSOperator, "LSR1": lo.LSR1Operator}[kind]
    op = make(torch.float64, n, mem=mem, device=dev)
    T = lambda a: torch.from_numpy(a).to(dev)
    pairs = []
    for _ in range(2 * mem + 3):
        s = rng.uniform(-1, 1, n)
        pairs.append((T(s), T(s * rng.uniform(0.5, 2.0, n) + (0.3 * rng.standard_normal(n) if kind == "LSR1" else 0))))
    for s, y in pairs[:mem + 1]:                                     # fill the memory (first pushes may grow workspaces)
        lo.push(op, s, y)
    for s, y in pairs[mem + 1:]:
        d = delta(lo, lambda: lo.push(op, s, y))
        assert d["malloc"] == 0 and d["free"] == 0 and d["memset_async"] <= 1, (kind, d)
        assert d["d2h"] == 1 and d["d2h_bytes"] <= 64, f"push! {kind}: exactly one small D2H copy, got {d}"
        assert d["h2d"] == 0 and d["blocking_copy"] == 0 and d["device_sync"] == 0, (kind, d)
        assert d["stream_sync"] + d["event_sync"] == 1, f"push! {kind}: one wait for that copy, got {d}"


def test_contract_also_holds_for_big_memory_layout_and_dense_leaves(lo, dev):
    rng = np.random.default_rng(2)
    n = 20_001
    T = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
    B = lo.LBFGSOperator(torch.float64, n, mem=100, device=dev)      # HBM-resident slot metadata (mem > 32)
    for _ in range(12):
        s = rng.uniform(-1, 1, n)
        lo.push(B, T(s), T(s * rng.uniform(0.5, 2.0, n)))
    v, res = T(rng.uniform(-1, 1, n)), torch.empty(n, dtype=torch.float64, device=dev)
    for _ in range(2):
        lo.mul(res, B, v, 1.0, 0.0)
    only_launches(delta(lo, lambda: lo.mul(res, B, v, 1.0, 0.0), 3), 3, "mul! LBFGS mem=100")
    m = 1500
    A = T(rng.standard_normal((m, m))).t()
    Hm = lo.opHermitian(T(rng.standard_normal(m)), A)
    M = lo.LinearOperatorFromMatrix(A)
    K = lo.kron(T(rng.standard_normal((30, 40))).t(), T(rng.standard_normal((50, 20))).t())     # (40x30) (x) (20x50)
    x, y = T(rng.uniform(-1, 1, m)), torch.empty(m, dtype=torch.float64, device=dev)
    xk, yk = T(rng.uniform(-1, 1, 30 * 50)), torch.empty(40 * 20, dtype=torch.float64, device=dev)
    for op, a, b, nm in ((Hm, x, y, "opHermitian"), (M, x, y, "LinearOperator(M)"), (M.T, x, y, "transpose(M)"),
                         (K, xk, yk, "kron")):
        for _ in range(2):
            lo.mul(b, op, a, 2.0, -1.0)
        only_launches(delta(lo, lambda: lo.mul(b, op, a, 2.0, -1.0), 3), 3, f"mul! {nm}")
    R = lo.opRestriction(np.sort(rng.choice(m, 700, replace=False)) + 1, m, device=dev)
    out = torch.empty(700, dtype=torch.float64, device=dev)
    lo.mul(out, R, x)
    only_launches(delta(lo, lambda: lo.mul(out, R, x), 2), 2, "mul! opRestriction")
    lo.mul(y, R.H, out)
    only_launches(delta(lo, lambda: lo.mul(y, R.H, out), 2), 2, "mul! opExtension (segment-owner kernel, no memset)")
