"""-m gpu: quasi-Newton operators through the C ABI vs the oracle.

Tolerances (fp64): reference-ordered inverse two-loop, forward L-BFGS, L-SR1: 1e-10 relative L2;
two-pass (Gram) inverse form: 1e-9; shifted solve (coefficient space): 1e-8 relative on
well-conditioned pairs. fp32: 2e-4. All far inside the reference's own sqrt(eps) ~ 1.5e-8 /
isapprox(1e-6) bars for the properties it tests."""
import numpy as np
import pytest
import torch

import oracle
from tolerances import QN_F32, QN_F32_ROUNDTRIP, observe

pytestmark = pytest.mark.gpu
NP = {torch.float64: np.float64, torch.float32: np.float32}
SV = lambda n: np.array([-(-1.0) ** i for i in range(1, n + 1)])


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def rel(a, b):
    f32 = getattr(a, "dtype", None) == np.float32
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    nb = np.linalg.norm(b)
    e = np.linalg.norm(a - b) / (nb if nb else 1.0)
    if f32:   # every Float32 comparison of this file feeds the observed envelope (tests/tolerances.py), keyed by test
        import os
        observe("by test: " + os.environ.get("PYTEST_CURRENT_TEST", "?").split("::")[-1].split(" ")[0].replace("[", " [", 1).split(" [")[0]
                + (" [lsr1]" if "lsr1" in os.environ.get("PYTEST_CURRENT_TEST", "") else ""), e)
    return e


def pairs(rng, n, k, dtype=np.float64):
    out = []
    for _ in range(k):
        s = rng.uniform(-1, 1, n)
        y = s * rng.uniform(0.5, 2.0, n) + 1e-2 * rng.standard_normal(n)
        out.append((s.astype(dtype), y.astype(dtype)))
    return out


# ------------------------------------------------------------------------------- KATs
def test_kat_solve_shifted_system(lo, dev, kat):
    """test/test_solve_shifted_system.jl:5-61 through the ABI: the exact rational solution of (B + σI) x = b for the dense
    BFGS matrix of the kept pairs (scaling off / on, σ = 0, 1/8, 2, 3), every forward push mode; result is x itself
    (`result === x_sol`), finite; negative σ raises; ldiv!(x, B, b) ≈ H*b with the inverse operator (σ = 0)."""
    cs = [c for c in kat if c["kind"] == "solve_shifted"]
    assert len(cs) == 5
    for c in cs:
        n = c["n"]
        for pm in ("compact", "gram", "reforder"):
            B = lo.LBFGSOperator(n, mem=c["mem"], scaling=c["scaling"], device=dev).set_push_mode(pm)
            H = lo.InverseLBFGSOperator(n, mem=c["mem"], scaling=False, device=dev)
            for p in c["pairs"]:
                lo.push(B, T(np.array(p["s"]), dev), T(np.array(p["y"]), dev))
                lo.push(H, T(np.array(p["s"]), dev), T(np.array(p["y"]), dev))
            b = T(np.array(c["b"]), dev)
            x_sol = torch.zeros(n, dtype=torch.float64, device=dev)
            result = lo.solve_shifted_system(x_sol, B, b, c["sigma"])
            assert result is x_sol and len(result) == n and bool(torch.isfinite(result).all())      # :30-37
            assert rel(x_sol.cpu().numpy(), np.array(c["expect_x"])) <= 1e-10, (c["name"], pm)
            assert np.allclose(x_sol.cpu().numpy(), np.array(c["x_true"]), atol=1e-6, rtol=1e-6)      # :40
            if "expect_Hb" in c:
                xl = lo.ldiv(torch.zeros(n, dtype=torch.float64, device=dev), B, b)                  # :49-60
                xH = (H * b).cpu().numpy()
                assert rel(xH, np.array(c["expect_Hb"])) <= 1e-10
                assert np.allclose(xl.cpu().numpy(), xH, atol=1e-6, rtol=1e-6)
            with pytest.raises(ValueError):
                lo.solve_shifted_system(x_sol, B, b, -0.1)                                           # :43-47


def test_kat_lbfgs(lo, dev, kat):
    """test_lbfgs.jl:7-70 through the ABI (values from dense BFGS in exact rationals)."""
    for c in [c for c in kat if c["kind"] == "lbfgs"]:
        n, mem = c["n"], c["mem"]
        B = lo.LBFGSOperator(n, mem=mem, scaling=c["scaling"], device=dev)
        H = lo.InverseLBFGSOperator(n, mem=mem, scaling=c["scaling"], device=dev)
        assert lo.isallocated5(B) and lo.isallocated5(H)
        I = np.eye(n)
        for t in range(2):                                       # run again after reset! (:13)
            assert np.array_equal(lo.Matrix(B).cpu().numpy(), I) and np.array_equal(lo.Matrix(H).cpu().numpy(), I)
            assert B.data.insert == 1 and H.data.insert == 1
            for p in c["pre_rejected"]:
                lo.push(B, T(np.array(p["s"]), dev), T(np.array(p["y"]), dev))
                lo.push(H, T(np.array(p["s"]), dev), T(np.array(p["y"]), dev))
                assert B.data.insert == 1 and H.data.insert == 1
            for p in c["pairs"]:
                lo.push(B, T(np.array(p["s"]), dev), T(np.array(p["y"]), dev))
                lo.push(H, T(np.array(p["s"]), dev), T(np.array(p["y"]), dev))
            assert B.data.insert == H.data.insert == c["expect_insert"]
            assert np.array_equal(B.data.ys, np.array(c["expect_ys_slots"]))
            v = T(np.array(c["v"]), dev)
            assert rel((B * v).cpu().numpy(), np.array(c["expect_Bv"])) <= 1e-12
            assert rel((H * v).cpu().numpy(), np.array(c["expect_Hv"])) <= 1e-12
            H.set_mode("reforder")
            assert rel((H * v).cpu().numpy(), np.array(c["expect_Hv"])) <= 1e-12
            H.set_mode("twopass")
            assert rel(lo.diag(B).cpu().numpy(), np.array(c["expect_diagB"])) <= 1e-12
            MB, MH = lo.Matrix(B).cpu().numpy(), lo.Matrix(H).cpu().numpy()
            assert np.linalg.norm(np.diag(MB) - lo.diag(B).cpu().numpy()) <= 1e-8           # :54
            assert np.linalg.norm(MH @ MB - I) <= np.sqrt(np.finfo(float).eps)              # :56
            assert np.allclose(MB, MB.T, rtol=0, atol=1e-12) and np.linalg.eigvalsh(MB).min() > 0
            assert np.linalg.norm(MB, 2) <= B.data.opnorm_upper_bound                        # :70
            assert rel((lo.compose(H, B) * v).cpu().numpy(), np.array(c["v"])) <= 1e-12
            lo.reset(B); lo.reset(H)                                                         # :62-67
            assert B.data.scaling_factor == 1.0 and H.data.scaling_factor == 1.0 and B.nprod == 0
            assert np.linalg.norm((B * v).cpu().numpy() - np.array(c["v"])) < 1e-8
    with pytest.raises(lo.LinearOperatorException):
        lo.diag(lo.InverseLBFGSOperator(5, device=dev))


def test_kat_lsr1(lo, dev, kat):
    """test_lsr1.jl:6-41."""
    for c in [c for c in kat if c["kind"] == "lsr1"]:
        n = c["n"]
        B = lo.LSR1Operator(n, mem=c["mem"], scaling=c["scaling"], device=dev)
        for t in range(2):
            assert np.array_equal(lo.Matrix(B).cpu().numpy(), np.eye(n)) and B.data.insert == 1
            s = T(SV(n), dev)
            lo.push(B, s, B * s)                                                    # rejected (:18-21)
            assert B.data.insert == 1
            nacc = 0
            for p in c["pairs"]:
                lo.push(B, T(np.array(p["s"]), dev), T(np.array(p["y"]), dev))
                nacc += B._last_push_accepted
            assert nacc == c["expect_naccepted"] and B.data.insert == c["expect_insert"]
            v = T(np.array(c["v"]), dev)
            assert rel((B * v).cpu().numpy(), np.array(c["expect_Bv"])) <= 1e-12
            assert rel(lo.diag(B).cpu().numpy(), np.array(c["expect_diagB"])) <= 1e-12
            MB = lo.Matrix(B).cpu().numpy()
            assert np.allclose(MB, MB.T, atol=1e-12) and np.linalg.norm(MB, 2) <= B.data.opnorm_upper_bound
            lo.reset(B)
            assert B.data.scaling_factor == 1.0


# ------------------------------------------------------------------------------- seeded parity
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("push_mode", ["gram", "reforder", "compact"])
@pytest.mark.parametrize("n,mem,npush,scaling", [(1000, 5, 8, True), (4097, 10, 13, True), (257, 3, 2, False),
                                                  (100_003, 7, 7, True), (64, 1, 3, True), (50_000, 20, 23, True),
                                                  (3001, 32, 40, True)])
def test_lbfgs_parity(lo, dev, dtype, push_mode, n, mem, npush, scaling):
    rng = np.random.default_rng(n + mem)
    npd = NP[dtype]
    tol = dict(ref=1e-10, two=1e-9, fwd=1e-10) if dtype == torch.float64 else dict(ref=QN_F32, two=QN_F32, fwd=QN_F32)
    if push_mode in ("gram", "compact"):   # Gram-form rebuild of the a_k panel: different association order
        tol["fwd"] = 1e-9 if dtype == torch.float64 else QN_F32
    B = lo.LBFGSOperator(dtype, n, mem=mem, scaling=scaling, device=dev).set_push_mode(push_mode)
    H = lo.InverseLBFGSOperator(dtype, n, mem=mem, scaling=scaling, device=dev)
    Bo = oracle.LBFGS(n, mem=mem, scaling=scaling, inverse=False, dtype=npd)
    Ho = oracle.LBFGS(n, mem=mem, scaling=scaling, inverse=True, dtype=npd)
    x, r0 = rng.uniform(-1, 1, n).astype(npd), rng.uniform(-1, 1, n).astype(npd)
    for k, (s, y) in enumerate(pairs(rng, n, npush, npd)):
        lo.push(B, T(s, dev), T(y, dev)); lo.push(H, T(s, dev), T(y, dev))
        Bo.push(s, y); Ho.push(s, y)
        assert B.data.insert == Bo.insert and H.data.insert == Ho.insert
        if k in (0, npush // 2, npush - 1):
            for alpha, beta in ((1.0, 0.0), (-1.0, 0.0), (2.0, -3.0)):
                fl = oracle.SCALARS_F64 if dtype == torch.float32 else 0
                res = T(r0.copy(), dev)
                if beta == 0:
                    res.fill_(float("nan"))
                lo.mul(res, B, T(x, dev), alpha, beta)
                e_ = observe(f"parity fwd mul! ({push_mode})", rel(res.cpu().numpy(), Bo.mul(r0.copy(), x, alpha, beta, flags=fl)), dtype == torch.float32)
                assert e_ <= tol["fwd"], ("fwd", k)
                want = Ho.mul(r0.copy(), x, alpha, beta, flags=fl)
                for mode in ("twopass", "reforder"):
                    H.set_mode(mode)
                    res = T(r0.copy(), dev)
                    if beta == 0:
                        res.fill_(float("nan"))
                    lo.mul(res, H, T(x, dev), alpha, beta)
                    e_ = observe(f"parity inv mul! ({mode})", rel(res.cpu().numpy(), want), dtype == torch.float32)
                    assert e_ <= tol["two" if mode == "twopass" else "ref"], (mode, k)
                H.set_mode("twopass")
    assert abs(B.data.scaling_factor - Bo.scaling_factor) <= 1e-6 * abs(Bo.scaling_factor)
    assert observe(f"parity fwd diag! ({push_mode})", rel(lo.diag(B).cpu().numpy(), Bo.diag()), dtype == torch.float32) <= tol["fwd"]
    assert rel(B.data.opnorm_upper_bound, Bo.opnorm_upper_bound) <= 1e-5
    # H*(B*x) == x (the property behind test_lbfgs.jl:56) at this size
    back = H * (B * T(x, dev))
    assert observe("parity H*(B*x) round trip", rel(back.cpu().numpy(), x), dtype == torch.float32) <= (1e-8 if dtype == torch.float64 else QN_F32_ROUNDTRIP)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("push_mode", ["gram", "reforder"])
@pytest.mark.parametrize("n,mem,npush,scaling", [(1000, 5, 8, True), (4099, 8, 11, False), (100_001, 6, 9, True),
                                                  (2001, 32, 37, True)])
def test_lsr1_parity(lo, dev, dtype, push_mode, n, mem, npush, scaling):
    rng = np.random.default_rng(7 * n + mem)
    npd = NP[dtype]
    tol = (1e-10 if push_mode == "reforder" else 1e-9) if dtype == torch.float64 else QN_F32
    B = lo.LSR1Operator(dtype, n, mem=mem, scaling=scaling, device=dev).set_push_mode(push_mode)
    Bo = oracle.LSR1(n, mem=mem, scaling=scaling, dtype=npd)
    x, r0 = rng.uniform(-1, 1, n).astype(npd), rng.uniform(-1, 1, n).astype(npd)
    for s, y in pairs(rng, n, npush, npd):
        lo.push(B, T(s, dev), T(y, dev))
        assert B._last_push_accepted == Bo.push(s, y)
        assert B.data.insert == Bo.insert
    for alpha, beta in ((1.0, 0.0), (2.0, -3.0), (-1, 1)):
        fl = oracle.SCALARS_F64 if (dtype == torch.float32 and isinstance(alpha, float)) else 0
        res = T(r0.copy(), dev)
        lo.mul(res, B, T(x, dev), alpha, beta)
        e_ = observe(f"parity lsr1 mul! ({push_mode})", rel(res.cpu().numpy(), Bo.mul(r0.copy(), x, float(alpha), float(beta), flags=fl)), dtype == torch.float32)
        assert e_ <= tol
    assert observe(f"parity lsr1 diag! ({push_mode})", rel(lo.diag(B).cpu().numpy(), Bo.diag()), dtype == torch.float32) <= tol
    assert rel(B.data.opnorm_upper_bound, Bo.opnorm_upper_bound) <= 1e-4


def test_lbfgs_vs_dense_bfgs(lo, dev):
    """test_lbfgs.jl:73-99 / test_lsr1.jl:43-68 with random pairs."""
    rng = np.random.default_rng(1)
    n = mem = 12
    LB = lo.LBFGSOperator(n, mem=mem, scaling=False, device=dev)
    LS = lo.LSR1Operator(n, mem=mem, scaling=False, device=dev)
    Bd, Sd = np.eye(n), np.eye(n)
    for _ in range(mem):
        s = rng.uniform(-1, 1, n)
        y = s * rng.uniform(0.5, 2.0, n)
        Bs = Bd @ s
        Bd = Bd - np.outer(Bs, Bs) / (s @ Bs) + np.outer(y, y) / (y @ s)
        r = y - Sd @ s
        if abs(r @ s) >= 1e-8 + 1e-8 * np.linalg.norm(s) * np.linalg.norm(r):
            Sd = Sd + np.outer(r, r) / (r @ s)
        lo.push(LB, T(s, dev), T(y, dev)); lo.push(LS, T(s, dev), T(y, dev))
        assert np.linalg.norm(lo.Matrix(LB).cpu().numpy() - Bd) < 1e-8 * np.linalg.norm(Bd)
        assert np.linalg.norm(lo.Matrix(LS).cpu().numpy() - Sd) < 1e-7 * np.linalg.norm(Sd)
        assert np.linalg.norm(lo.diag(LB).cpu().numpy() - np.diag(Bd)) < 1e-8 * np.linalg.norm(np.diag(Bd))


def test_damped_pushes(lo, dev):
    """test_lbfgs.jl:104-159 (same loop: d = -(H*g), s = α d, push when ys > 0.2 s'Bs) through the ABI and
    through the oracle, plus the misuse errors of :220-240."""
    n, mem = 10, 5
    B = lo.LBFGSOperator(n, mem=mem, damped=True, scaling=False, sigma2=0.8, sigma3=float("inf"), device=dev)
    H = lo.InverseLBFGSOperator(n, mem=mem, damped=True, scaling=False, sigma2=0.8, sigma3=float("inf"), device=dev)
    Bo = oracle.LBFGS(n, mem=mem, damped=True, scaling=False, inverse=False, sigma2=0.8, sigma3=np.inf)
    Ho = oracle.LBFGS(n, mem=mem, damped=True, scaling=False, inverse=True, sigma2=0.8, sigma3=np.inf)
    ins = 0
    for i in range(1, mem + 3):
        y = SV(n)
        ys = y @ SV(n)
        g = SV(n)
        d = -(H * T(g, dev))
        alpha = i / mem
        s = alpha * d
        do = -Ho.mul(np.empty(n), g)
        so = alpha * do
        assert rel(s.cpu().numpy(), so) <= 1e-10
        if ys > 0.2 * float(torch.dot(s, B * s)):
            ins += 1
            yd = T(y, dev)
            lo.push(B, s, T(y, dev)); lo.push(H, s, yd, alpha, T(g, dev))
            yo = y.copy()
            Bo.push(so.copy(), y.copy()); Ho.push(so.copy(), yo, alpha=alpha, g=g)
            assert rel(yd.cpu().numpy(), yo) <= 1e-10       # damping mutates y in place (lbfgs.jl:351)
    assert ins > 0 and B.data.insert == ins % mem + 1 and H.data.insert == ins % mem + 1
    assert B.data.insert == Bo.insert and H.data.insert == Ho.insert
    MB, MH = lo.Matrix(B).cpu().numpy(), lo.Matrix(H).cpu().numpy()
    assert np.linalg.eigvalsh((MB + MB.T) / 2).min() > 0 and np.linalg.eigvalsh((MH + MH.T) / 2).min() > 0
    assert np.allclose(MB, MB.T, atol=1e-10) and np.allclose(MH, MH.T, atol=1e-10)
    assert np.linalg.norm(np.diag(MB) - lo.diag(B).cpu().numpy()) <= 1e-8
    assert np.linalg.norm(MH @ MB - np.eye(n)) <= np.sqrt(np.finfo(float).eps)
    assert np.linalg.norm(MB, 2) <= B.data.opnorm_upper_bound
    assert rel(MB, Bo.dense()) <= 1e-9 and rel(MH, Ho.dense()) <= 1e-9
    # full-memory damped L-BFGS vs dense damped BFGS (:143-156)
    LB = lo.LBFGSOperator(n, mem=n, damped=True, scaling=False, device=dev)
    Bd = np.eye(n)
    for k in range(n):
        s, y = SV(n), SV(n)
        Bs = Bd @ s
        if y @ s > 0.2 * (s @ Bs):
            Bd = Bd - np.outer(Bs, Bs) / (s @ Bs) + np.outer(y, y) / (y @ s)
        lo.push(LB, T(s, dev), T(y, dev))
        assert np.linalg.norm(lo.Matrix(LB).cpu().numpy() - Bd) < 1e-8 * np.linalg.norm(Bd)
    x = T(SV(n), dev)
    U = lo.LBFGSOperator(n, device=dev)
    with pytest.raises(RuntimeError):
        lo.push(U, x, x, x)
    with pytest.raises(RuntimeError):
        lo.push(B, x, x, 1.0, x)
    with pytest.raises(RuntimeError):
        lo.push(H, x, x, x)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("scaling", [False, True])
def test_solve_shifted_system(lo, dev, dtype, scaling):
    """test_solve_shifted_system.jl:5-61: build b = Bx + σx, recover x; ldiv! == H*b; σ<0 -> error."""
    rng = np.random.default_rng(3)
    n, M = 100, 5
    npd = NP[dtype]
    B = lo.LBFGSOperator(dtype, n, mem=M, scaling=scaling, device=dev)
    H = lo.InverseLBFGSOperator(dtype, n, mem=M, scaling=scaling, device=dev)
    Bo = oracle.LBFGS(n, mem=M, scaling=scaling, inverse=False, dtype=npd)
    for _ in range(10):
        s, y = rng.random(n).astype(npd), rng.random(n).astype(npd)
        lo.push(B, T(s, dev), T(y, dev)); lo.push(H, T(s, dev), T(y, dev)); Bo.push(s, y)
    x = rng.standard_normal(n).astype(npd)
    at = 1e-6 if dtype == torch.float64 else QN_F32_ROUNDTRIP
    for sigma in (0.1, 0.0, 3.0):
        b = (B * T(x, dev)) + sigma * T(x, dev)
        xs = torch.zeros(n, dtype=dtype, device=dev)
        out = lo.solve_shifted_system(xs, B, b, sigma)
        assert out is xs and torch.isfinite(xs).all()
        observe("solve_shifted_system! round trip (max abs/|x| err)", np.abs(xs.cpu().numpy() - x).max() / (np.abs(x).max() + 1e-300), dtype == torch.float32)
        assert np.allclose(xs.cpu().numpy(), x, atol=at, rtol=at)
        if dtype == torch.float64:   # against the oracle's statement-by-statement recursion
            want = Bo.solve_shifted(np.zeros(n), b.cpu().numpy(), sigma)
            assert rel(xs.cpu().numpy(), want) <= 1e-8
    b = B * T(x, dev)
    xs = lo.ldiv(torch.zeros(n, dtype=dtype, device=dev), B, b)
    assert np.allclose(xs.cpu().numpy(), (H * b).cpu().numpy(), atol=at, rtol=at)
    with pytest.raises(ValueError):
        lo.solve_shifted_system(xs, B, b, -0.1)
    # partially filled memory and empty operator
    B2 = lo.LBFGSOperator(dtype, n, mem=8, scaling=scaling, device=dev)
    xs = lo.solve_shifted_system(torch.zeros(n, dtype=dtype, device=dev), B2, T(x, dev), 1.0)
    assert np.allclose(xs.cpu().numpy(), x / 2, rtol=1e-6)
    lo.push(B2, T(rng.random(n).astype(npd), dev), T(rng.random(n).astype(npd), dev))
    b = (B2 * T(x, dev)) + 0.5 * T(x, dev)
    xs = lo.solve_shifted_system(torch.zeros(n, dtype=dtype, device=dev), B2, b, 0.5)
    assert np.allclose(xs.cpu().numpy(), x, atol=at, rtol=at)


def test_lbfgs_bench_size_properties(lo, dev):
    """BASELINE config 3 size (m=10, n=5e7): H*(B*x) == x, linearity, determinism; reference-ordered and
    two-pass inverse forms agree."""
    n, m = 50_000_000, 10
    g = torch.Generator(device=dev).manual_seed(11)
    B = lo.LBFGSOperator(torch.float64, n, mem=m, device=dev)
    H = lo.InverseLBFGSOperator(torch.float64, n, mem=m, device=dev)
    for _ in range(m + 3):
        s = torch.rand(n, dtype=torch.float64, device=dev, generator=g) * 2 - 1
        y = (torch.rand(n, dtype=torch.float64, device=dev, generator=g) * 1.5 + 0.5) * s
        y += 1e-2 * (torch.rand(n, dtype=torch.float64, device=dev, generator=g) - 0.5)
        lo.push(B, s, y); lo.push(H, s, y)
        del s, y
    x = torch.rand(n, dtype=torch.float64, device=dev, generator=g) * 2 - 1
    hx = H * x
    assert torch.equal(hx, H * x)                                           # deterministic
    H.set_mode("reforder")
    hx_ref = H * x
    H.set_mode("twopass")
    assert (torch.linalg.vector_norm(hx - hx_ref) / torch.linalg.vector_norm(hx_ref)).item() <= 1e-9
    del hx_ref
    back = B * hx
    assert (torch.linalg.vector_norm(back - x) / torch.linalg.vector_norm(x)).item() <= 1e-8
    del back
    out = torch.empty_like(x)
    lo.mul(out, H, x, 2.0, 0.0)
    assert (torch.linalg.vector_norm(out - 2 * hx) / torch.linalg.vector_norm(hx)).item() <= 1e-14


def test_allreduce_hook_world1(lo, dev):
    """The row-sharding hook is invoked on every global reduction; with world_size == 1 the RCCL
    all-reduce is the identity, so results must be unchanged (multi-rank logic: tests/test_sharded_cpu.py)."""
    import os
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        ctx = lo.get_ctx(dev)
        calls = []
        inner = lo.sharded.make_allreduce_hook(None, cuda=True)

        def hook(user, buf, count, stream):
            calls.append(int(count))
            return inner(user, buf, count, stream)

        rng = np.random.default_rng(0)
        n = 10_000
        h = rng.standard_normal(n); h /= np.linalg.norm(h)
        v = rng.uniform(-1, 1, n)
        H = lo.opHouseholder(T(h, dev))
        ctx.tune("house_fused", 0)               # a hooked apply is the two-launch path (the hook sits between the passes):
        want = (H * T(v, dev)).clone()           # compare with the SAME reduction order, un-hooked
        ctx.tune("house_fused", 1)
        ctx.set_allreduce(hook)
        got = H * T(v, dev)
        assert torch.equal(got, want) and calls == [1]
        Hq = lo.InverseLBFGSOperator(n, mem=4, device=dev)
        for s, y in pairs(rng, n, 5):
            lo.push(Hq, T(s, dev), T(y, dev))
        calls.clear()
        r1 = Hq * T(v, dev)
        assert calls == [8]                      # ONE all-reduce of 2m doubles per apply
        ctx.set_allreduce(None)
        ctx.tune("qn_fused_small", 0)            # a hooked apply is the four-launch schedule: same reduction order un-hooked
        try:
            assert torch.equal(r1, Hq * T(v, dev))
        finally:
            ctx.tune("qn_fused_small", 1)
    finally:
        lo.get_ctx(dev).set_allreduce(None)
        dist.destroy_process_group()


def test_forward_lbfgs_shard_size_properties(lo, dev):
    """BASELINE config 5 per-GPU shard (LBFGSOperator m=20, n_local=5e7): linearity of the apply, the shifted
    solve inverts (B + σI) at this size, diag! matches probing with unit-like vectors on a sample."""
    n, m = 50_000_000, 20
    g = torch.Generator(device=dev).manual_seed(13)
    B = lo.LBFGSOperator(torch.float64, n, mem=m, device=dev)
    for _ in range(m + 2):
        s = torch.rand(n, dtype=torch.float64, device=dev, generator=g) * 2 - 1
        y = (torch.rand(n, dtype=torch.float64, device=dev, generator=g) * 1.5 + 0.5) * s
        lo.push(B, s, y)
        del s, y
    assert B.data.insert == (m + 2) % m + 1
    x = torch.rand(n, dtype=torch.float64, device=dev, generator=g) * 2 - 1
    bx = B * x
    out = torch.empty_like(x)
    lo.mul(out, B, x, -2.0, 0.0)
    assert (torch.linalg.vector_norm(out + 2 * bx) / torch.linalg.vector_norm(bx)).item() <= 1e-14
    sigma = 0.25
    rhs = bx + sigma * x
    sol = lo.solve_shifted_system(torch.zeros_like(x), B, rhs, sigma)
    assert (torch.linalg.vector_norm(sol - x) / torch.linalg.vector_norm(x)).item() <= 1e-8
    del rhs, sol, out
    d = lo.diag(B)
    assert torch.isfinite(d).all() and (d > 0).all()        # B is SPD: positive diagonal


def test_native_rccl_hook_world1(lo, dev):
    """libmxlo_rccl.so: a 1-rank RCCL communicator, `mxlo_rccl_allreduce_hook` installed as the ctx hook
    (ncclAllReduce issued from C on the ctx stream). With one rank the sum is the identity: results must
    be bit-identical to the un-hooked run, for Householder and for an L-BFGS apply + push."""
    torch.cuda.set_device(dev)
    ctx = lo.get_ctx(dev)
    rng = np.random.default_rng(5)
    n = 100_003
    h = rng.standard_normal(n); h /= np.linalg.norm(h)
    v = rng.uniform(-1, 1, n)
    H = lo.opHouseholder(T(h, dev))
    ctx.tune("house_fused", 0)                   # same (two-launch) reduction order as the hooked apply
    want = (H * T(v, dev)).clone()
    ctx.tune("house_fused", 1)
    hook = lo.sharded.NativeRcclHook(0, 1)
    try:
        hook.install(ctx)
        assert torch.equal(H * T(v, dev), want)
        B = lo.LBFGSOperator(n, mem=4, device=dev)
        for s, y in pairs(rng, n, 6):
            lo.push(B, T(s, dev), T(y, dev))
        r1 = B * T(v, dev)
        ctx.set_allreduce(None)
        ctx.tune("qn_fused_small", 0)            # same (four-launch) reduction order as the hooked apply
        assert torch.equal(r1, B * T(v, dev))
    finally:
        ctx.tune("qn_fused_small", 1)
        ctx.set_allreduce(None)
        torch.cuda.synchronize()
        hook.close()


# ------------------------------------------------------------------------------- fused ShiftedOperator
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("kind", ["fwd", "inv", "inv_reforder", "lsr1"])
@pytest.mark.parametrize("n", [1, 1001, 65_537])
def test_shifted_qn_fused_is_bit_identical(lo, dev, dtype, kind, n):
    """ShiftedOperator(H, σ) over a quasi-Newton H (src/shifted_operators.jl:16-25): the fused apply
    (mxlo_qn_mul_shifted: axpy! folded into the combine pass) must reproduce, bit for bit, mul!(y,H,x,α,β)
    followed by axpy!(ασ, x, y) — for every operator kind, Float64 and Float32 scalars, β = 0 and β ≠ 0 —
    and agree with the oracle's H plus σI."""
    npd = NP[dtype]
    rng = np.random.default_rng(n + len(kind))
    mem = 4
    if kind == "fwd":
        op, orc = lo.LBFGSOperator(dtype, n, mem=mem, device=dev), oracle.LBFGS(n, mem=mem, inverse=False, dtype=npd)
    elif kind == "lsr1":
        op, orc = lo.LSR1Operator(dtype, n, mem=mem, device=dev), oracle.LSR1(n, mem=mem, dtype=npd)
    else:
        op, orc = lo.InverseLBFGSOperator(dtype, n, mem=mem, device=dev), oracle.LBFGS(n, mem=mem, inverse=True, dtype=npd)
        if kind == "inv_reforder":
            op.set_mode("reforder")
    for s, y in pairs(rng, n, mem + 2, npd):
        lo.push(op, T(s, dev), T(y, dev))
        orc.push(s, y)
    x, r0 = rng.uniform(-1, 1, n).astype(npd), rng.uniform(-1, 1, n).astype(npd)
    sigma = 0.37
    Sh = lo.ShiftedOperator(op, sigma)
    f32s = (np.float32(1.5), np.float32(-0.25))
    for a, b in [(1.0, 0.0), (2.0, -3.0)] + ([f32s] if dtype == torch.float32 else []):
        outs = []
        for fuse in (True, False):
            type(Sh).fuse = fuse
            try:
                res = T(r0.copy(), dev) if b != 0 else torch.full((n,), float("nan"), dtype=dtype, device=dev)
                lo.mul(res, Sh, T(x, dev), a, b)
                outs.append(res)
            finally:
                type(Sh).fuse = True
        assert torch.equal(outs[0], outs[1]), (kind, a, b)
        ref = orc.mul(r0.copy(), x, float(a), float(b)).astype(np.float64) + float(a) * sigma * x.astype(np.float64)
        assert rel(outs[0].cpu().numpy(), ref) <= (1e-9 if dtype == torch.float64 else QN_F32)
    nb = lo.nprod(op)
    Sh.data.sigma = 0.0                                   # σ == 0 (or α == 0): plain mul!, no axpy (:21)
    assert torch.equal(Sh * T(x, dev), op * T(x, dev)) and lo.nprod(op) == nb + 2


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_compact_forward_push_materialises_on_demand(lo, dev, dtype):
    """MXLO_PUSH_COMPACT: push! leaves a_k = [S B]·c_k implicit and mul! works on [S B]; diag!, solve_shifted_system!,
    the panel accessor and a later reference-ordered push! must see the same operator as the materialising modes."""
    npd = NP[dtype]
    n, mem = 20_011, 6
    rng = np.random.default_rng(17)
    tol = 1e-9 if dtype == torch.float64 else QN_F32
    Bc = lo.LBFGSOperator(dtype, n, mem=mem, device=dev).set_push_mode("compact")
    Bg = lo.LBFGSOperator(dtype, n, mem=mem, device=dev).set_push_mode("gram")
    O = oracle.LBFGS(n, mem=mem, inverse=False, dtype=npd)
    x, b = rng.uniform(-1, 1, n).astype(npd), rng.uniform(-1, 1, n).astype(npd)
    ps = pairs(rng, n, mem + 3, npd)
    for k, (s, y) in enumerate(ps):
        lo.push(Bc, T(s, dev), T(y, dev)); lo.push(Bg, T(s, dev), T(y, dev)); O.push(s, y)
        want = O.mul(np.empty(n, npd), x)
        assert rel((Bc * T(x, dev)).cpu().numpy(), want) <= tol, k          # compact apply, a_k implicit
        if k == 2:                                                           # partially filled memory
            assert rel(lo.diag(Bc).cpu().numpy(), O.diag()) <= tol           # materialises once
            assert rel((Bc * T(x, dev)).cpu().numpy(), want) <= tol          # classic apply on the materialised panel
    assert rel((Bc * T(x, dev)).cpu().numpy(), (Bg * T(x, dev)).cpu().numpy()) <= tol
    xs = lo.solve_shifted_system(torch.zeros(n, dtype=dtype, device=dev), Bc, T(b, dev), npd(0.25))
    xg = lo.solve_shifted_system(torch.zeros(n, dtype=dtype, device=dev), Bg, T(b, dev), npd(0.25))
    assert rel(xs.cpu().numpy(), xg.cpu().numpy()) <= 10 * tol
    back = lo.ShiftedOperator(Bc, 0.25) * xs                                 # (B + σI) x = b
    assert rel(back.cpu().numpy(), b) <= (1e-8 if dtype == torch.float64 else 5e-3)
    # switch to the reference-ordered rebuild mid-stream: it reads the older a_l, which must exist by then
    s, y = pairs(rng, n, 1, npd)[0]
    Bc.set_push_mode("compact"); lo.push(Bc, T(s, dev), T(y, dev)); O.push(s, y)
    s, y = pairs(rng, n, 1, npd)[0]
    Bc.set_push_mode("reforder"); lo.push(Bc, T(s, dev), T(y, dev)); O.push(s, y)
    assert rel((Bc * T(x, dev)).cpu().numpy(), O.mul(np.empty(n, npd), x)) <= tol
    lo.reset(Bc)
    assert torch.equal(Bc * T(x, dev), T(x, dev))


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("mem,npush", [(33, 20), (48, 60), (64, 70)])
def test_inverse_lbfgs_large_memory(lo, dev, dtype, mem, npush):
    """The inverse operator supports mem up to 64 (128 panel columns per combine); forward L-BFGS and L-SR1 stop at 32
    (their coefficient kernels own one lane per basis vector) and say so."""
    npd = NP[dtype]
    n = 6007
    rng = np.random.default_rng(mem)
    H = lo.InverseLBFGSOperator(dtype, n, mem=mem, device=dev)
    Ho = oracle.LBFGS(n, mem=mem, inverse=True, dtype=npd)
    x, r0 = rng.uniform(-1, 1, n).astype(npd), rng.uniform(-1, 1, n).astype(npd)
    tol = dict(twopass=1e-9, reforder=1e-10) if dtype == torch.float64 else dict(twopass=QN_F32, reforder=QN_F32)
    for k, (s, y) in enumerate(pairs(rng, n, npush, npd)):
        lo.push(H, T(s, dev), T(y, dev)); Ho.push(s, y)
        if k in (3, npush // 2, npush - 1):
            assert H.data.insert == Ho.insert
            fl = oracle.SCALARS_F64 if dtype == torch.float32 else 0
            want = Ho.mul(r0.copy(), x, 2.0, -3.0, flags=fl)
            for mode in ("twopass", "reforder"):
                H.set_mode(mode)
                res = T(r0.copy(), dev)
                lo.mul(res, H, T(x, dev), 2.0, -3.0)
                assert rel(res.cpu().numpy(), want) <= tol[mode], (mode, k)
            H.set_mode("twopass")
    assert lo.LBFGSOperator(dtype, n, mem=32, device=dev).mem == 32


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("kind,mem,npush", [("fwd", 100, 130), ("fwd", 33, 20), ("inv", 100, 130), ("inv", 65, 40),
                                            ("lsr1", 40, 55), ("lsr1", 100, 60)])
def test_unbounded_memory(lo, dev, dtype, kind, mem, npush):
    """The reference accepts any `mem` (src/lbfgs.jl:26-57, src/lsr1.jl:19-34). Beyond the single-wave coefficient
    kernels (64 slots inverse, 32 forward / L-SR1) the device-resident path of csrc/qn_big.h takes over: same parity
    bars against the oracle for mul! (5-arg), push! incl. buffer wrap-around and rejections, diag!, reset! and — forward
    — solve_shifted_system!."""
    npd = NP[dtype]
    n = 3001
    rng = np.random.default_rng(1000 * mem + npush)
    f64 = dtype == torch.float64
    tol = (1e-9 if f64 else QN_F32)
    if kind == "lsr1":
        B = lo.LSR1Operator(dtype, n, mem=mem, device=dev)
        Bo = oracle.LSR1(n, mem=mem, scaling=True, dtype=npd)
    else:
        ctor = lo.LBFGSOperator if kind == "fwd" else lo.InverseLBFGSOperator
        B = ctor(dtype, n, mem=mem, device=dev)
        Bo = oracle.LBFGS(n, mem=mem, scaling=True, inverse=(kind == "inv"), dtype=npd)
    assert B.mem == mem
    x, r0 = rng.uniform(-1, 1, n).astype(npd), rng.uniform(-1, 1, n).astype(npd)
    checks = {0, 5, mem - 1, mem, npush - 1}
    for k, (s, y) in enumerate(pairs(rng, n, npush, npd)):
        if k == 7:                                         # a rejected pair (y's <= eps) must not disturb the state
            lo.push(B, T(s, dev), T(-s, dev))
            Bo.push(s, -s)
        lo.push(B, T(s, dev), T(y, dev))
        Bo.push(s, y)
        if k in checks:
            assert B.data.insert == Bo.insert, k
            for alpha, beta in ((1.0, 0.0), (2.0, -3.0)):
                fl = oracle.scalar_flags(npd, alpha, beta)
                res = T(r0.copy(), dev)
                if beta == 0:
                    res.fill_(float("nan"))
                lo.mul(res, B, T(x, dev), alpha, beta)
                assert rel(res.cpu().numpy(), Bo.mul(r0.copy(), x, alpha, beta, flags=fl)) <= tol, (kind, k, alpha)
            if kind == "inv":
                B.set_mode("reforder")
                res = T(r0.copy(), dev)
                lo.mul(res, B, T(x, dev), 2.0, -3.0)
                assert rel(res.cpu().numpy(), Bo.mul(r0.copy(), x, 2.0, -3.0, flags=oracle.scalar_flags(npd, 2.0, -3.0))) <= (1e-10 if f64 else QN_F32)
                B.set_mode("twopass")
    assert abs(B.data.scaling_factor - Bo.scaling_factor) <= 1e-6 * abs(Bo.scaling_factor)
    if kind != "inv":
        assert rel(lo.diag(B).cpu().numpy(), Bo.diag()) <= tol
        assert rel(B.data.opnorm_upper_bound, Bo.opnorm_upper_bound) <= 1e-4
    if kind == "fwd":
        sigma = 0.3
        b = (B * T(x, dev)) + sigma * T(x, dev)
        xs = lo.solve_shifted_system(torch.zeros(n, dtype=dtype, device=dev), B, b, sigma)
        assert np.allclose(xs.cpu().numpy(), x, atol=1e-6 if f64 else QN_F32_ROUNDTRIP, rtol=1e-6 if f64 else QN_F32_ROUNDTRIP)
        if f64:
            assert rel(xs.cpu().numpy(), Bo.solve_shifted(np.zeros(n), b.cpu().numpy(), sigma)) <= 1e-7
        with pytest.raises(lo.MxloError):
            B.set_push_mode("reforder")                   # a validation mode of the small-memory path only
    lo.reset(B)
    assert rel((B * T(x, dev)).cpu().numpy(), x) == 0.0   # empty memory: the identity
    s, y = pairs(rng, n, 1, npd)[0]
    lo.push(B, T(s, dev), T(y, dev))
    Bo2 = (oracle.LSR1(n, mem=mem, scaling=True, dtype=npd) if kind == "lsr1"
           else oracle.LBFGS(n, mem=mem, scaling=True, inverse=(kind == "inv"), dtype=npd))
    Bo2.push(s, y)
    assert rel((B * T(x, dev)).cpu().numpy(), Bo2.mul(np.empty(n, npd), x, 1.0, 0.0, flags=oracle.scalar_flags(npd, 1.0, 0.0))) <= tol


def test_mem_limit_message(lo, dev):
    with pytest.raises(lo.MxloError, match="exceeds 4096"):
        lo.LBFGSOperator(torch.float64, 8, mem=5000, device=dev)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("kind,mem", [("inv", 4), ("inv", 10), ("inv", 23), ("fwd", 5), ("fwd", 20), ("fwd", 32)])
def test_one_pass_push_matches_two_kernel_schedule_and_oracle(lo, dev, dtype, kind, mem):
    """push!(op, s, y) as two streaming passes with the new pair held per lane (the column being replaced
    is never read, the inserts s -> S[:,k], y -> Y[:,k], b = y ./ sqrt(ys) happen from registers, y's and y'y come out
    of the first pass) against the schedule it replaced (`push_fused` = 0: dots, two device-to-device inserts, dual-x
    dots over the freshly copied columns) and against the oracle (src/lbfgs.jl:210-287). Covers: ragged n (partial last
    vector: the panel padding must stay zero), a rejected pair (y's <= eps: the operator must be untouched), wrap-around
    of the circular buffer, more than 10 columns (two chunks per panel), misaligned s / y views (fall back)."""
    from linearoperators_jl_amd.device import get_ctx
    ctx = get_ctx(dev)
    npd = NP[dtype]
    n = 10_007 if mem < 20 else 6_001
    rng = np.random.default_rng(mem * 7 + n)
    make = lo.InverseLBFGSOperator if kind == "inv" else lo.LBFGSOperator
    tol = 1e-9 if dtype == torch.float64 else QN_F32
    ops = {}
    try:
        for fused in (1, 0):
            ctx.tune("push_fused", fused)
            ops[fused] = make(dtype, n, mem=mem, scaling=True, device=dev)
        Oo = oracle.LBFGS(n, mem=mem, scaling=True, inverse=(kind == "inv"), dtype=npd)
        x = rng.uniform(-1, 1, n).astype(npd)
        prs = pairs(rng, n, mem + 4, npd)
        prs.insert(3, (prs[0][0], (-prs[0][0]).astype(npd)))          # y's < 0: rejected (src/lbfgs.jl:281-284)
        for k, (s, y) in enumerate(prs):
            want = Oo.push(s, y)
            for fused in (1, 0):
                ctx.tune("push_fused", fused)
                if k % 5 == 4:                                         # misaligned views: element offset 1
                    sb, yb = torch.empty(n + 1, dtype=dtype, device=dev), torch.empty(n + 1, dtype=dtype, device=dev)
                    sb[1:].copy_(T(s, dev)); yb[1:].copy_(T(y, dev))
                    lo.push(ops[fused], sb[1:], yb[1:])
                else:
                    lo.push(ops[fused], T(s, dev), T(y, dev))
                assert ops[fused].data.insert == Oo.insert, (k, fused, want)
            if k in (0, 3, mem - 1, mem, len(prs) - 1):
                got = {}
                for fused in (1, 0):
                    res = torch.full((n,), float("nan"), dtype=dtype, device=dev)
                    lo.mul(res, ops[fused], T(x, dev), 1.0, 0.0)
                    got[fused] = res.cpu().numpy()
                ref = Oo.mul(np.empty(n, dtype=npd), x, 1.0, 0.0)
                assert rel(got[1], ref) <= tol and rel(got[0], ref) <= tol, (k, rel(got[1], ref), rel(got[0], ref))
                assert rel(got[1], got[0]) <= (1e-12 if dtype == torch.float64 else 1e-4), (k, rel(got[1], got[0]))
        for fused in (1, 0):                                           # replicated scalars of the two schedules
            assert abs(ops[fused].data.scaling_factor - Oo.scaling_factor) <= 1e-6 * abs(Oo.scaling_factor)
        if kind == "fwd":
            d1, d0 = lo.diag(ops[1]).cpu().numpy(), lo.diag(ops[0]).cpu().numpy()
            assert rel(d1, Oo.diag()) <= tol and rel(d1, d0) <= (1e-12 if dtype == torch.float64 else 1e-4)
            assert rel(ops[1].data.opnorm_upper_bound, Oo.opnorm_upper_bound) <= 1e-5
    finally:
        ctx.tune("push_fused", 1)


@pytest.mark.parametrize("kind", ["inv", "fwd", "lsr1"])
@pytest.mark.parametrize("n", [1, 4099, 3_000_017])
def test_posted_read_back_of_the_push_decision(lo, dev, kind, n):
    """push!'s few doubles of decision state (y's, y'y, ... — src/lbfgs.jl:281-284, src/lsr1.jl:131-141) come back
    through mapped pinned host memory that a one-wave kernel writes and the host polls (`push_posted` = 1), or through
    hipMemcpyAsync + a stream synchronisation (0). Same kernels, same numbers: every decision (incl. rejected pairs),
    the insert pointer, the scaling factor and every apply must be IDENTICAL; n = 3·10⁶ with mem = 20 makes the first
    pass outlast the polling window, i.e. runs the hand-over to the stream synchronisation. `push_posted` = 2 (debug)
    treats every posting as lost: the doubles are then copied from device memory and posting is switched off for that
    handle — what a platform whose mapped host memory does not see device stores would get instead of an error."""
    from linearoperators_jl_amd.device import get_ctx
    ctx = get_ctx(dev)
    mem = 20 if n > 10**6 else 5
    rng = np.random.default_rng(n + len(kind))
    make = {"inv": lo.InverseLBFGSOperator, "fwd": lo.LBFGSOperator, "lsr1": lo.LSR1Operator}[kind]
    ops = {}
    try:
        for posted in (1, 0, 2):
            ops[posted] = make(torch.float64, n, mem=mem, scaling=True, device=dev)
        x = T(rng.uniform(-1, 1, n), dev)
        prs = pairs(rng, n, 7 if n > 10**6 else mem + 4, np.float64)
        prs.insert(2, (prs[0][0], -prs[0][0]))                           # y's < 0: rejected by the L-BFGS operators
        prs.insert(4, (prs[1][0], np.zeros(n)))                          # y = 0: rejected by all three
        accepted = 0
        for k, (s, y) in enumerate(prs):
            got = {}
            for posted in (1, 0, 2):
                ctx.tune("push_posted", posted)
                lo.push(ops[posted], T(s, dev), T(y, dev))
                res = torch.full((n,), float("nan"), dtype=torch.float64, device=dev)
                lo.mul(res, ops[posted], x, 1.0, 0.0)
                got[posted] = res.cpu().numpy()
            for posted in (1, 2):
                assert ops[posted].data.insert == ops[0].data.insert, (k, posted)
                assert ops[posted].data.scaling_factor == ops[0].data.scaling_factor, (k, posted)
                assert np.array_equal(got[posted], got[0], equal_nan=True), (k, posted)
            assert n == 1 or np.isfinite(got[1]).all(), k
            accepted += int(getattr(ops[1], "_last_push_accepted", True))
        assert n == 1 or 3 <= accepted <= len(prs) - (0 if kind == "lsr1" else 2)   # both outcomes occurred (L-BFGS)
    finally:
        ctx.tune("push_posted", 1)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("mem,scaling", [(1, True), (4, True), (4, False), (10, True), (13, False), (25, True)])
def test_lsr1_streaming_push_matches_apply_based_schedule_and_oracle(lo, dev, dtype, mem, scaling):
    """push!(op::LSR1Operator, s, y) (src/lsr1.jl:119-184) as the streaming schedule — S and Y panels once against
    (s, y) for the Gram rows and, through the last rebuild's coefficients, the a_k's of B s; the a_k panel once for
    r = y - B s, never stored, with r's, |r|^2 and |y - s/sf|^2 out of the same pass; the rebuild A = [Y S] C reading the
    pair from the caller's vectors and dropping it into its slots — against the schedule it replaced (`push_fused` = 0: an
    apply, five dots, two inserts, dual dots, rebuild) and against the oracle. Covers: odd n (the inserts cannot ride in
    the rebuild), a partial last vector, rejected pairs of each kind (:131 y = B s on the empty memory; with scaling :137
    y ⟂ s and :141 y ∥ s), wrap-around, more than 10 / 20 columns, misaligned views (fall back), reset! and reuse."""
    from linearoperators_jl_amd.device import get_ctx
    ctx = get_ctx(dev)
    npd = NP[dtype]
    tol = 1e-9 if dtype == torch.float64 else QN_F32
    same = 1e-11 if dtype == torch.float64 else 1e-4
    try:
        for n in ((10_006, 6_001) if mem <= 10 else (4_002,)):
            rng = np.random.default_rng(mem * 11 + n)
            ops = {}
            for fused in (1, 0):
                ctx.tune("push_fused", fused)
                ops[fused] = lo.LSR1Operator(dtype, n, mem=mem, scaling=scaling, device=dev)
            Oo = oracle.LSR1(n, mem=mem, scaling=scaling, dtype=npd)
            x = rng.uniform(-1, 1, n).astype(npd)
            prs = pairs(rng, n, mem + 4, npd)
            s0 = prs[1][0]
            prs.insert(0, (s0, s0.copy()))                 # empty memory, B = I: y - B s = 0 exactly, not well defined (:131)
            prs.insert(3, (s0, (npd(3.0) * s0).astype(npd)))                     # y ∥ s: |y - s/sf| ≈ eps|y| ≪ eps|y||s| (:141)
            if scaling:
                yperp = np.zeros(n, npd); yperp[0], yperp[1] = s0[1], -s0[0]     # y's = O(eps): no curvature (:137)
                prs.insert(5, (s0, yperp))
            nacc = 0
            for k, (s, y) in enumerate(prs):
                want = Oo.push(s, y)
                nacc += want
                for fused in (1, 0):
                    ctx.tune("push_fused", fused)
                    if k % 5 == 4:                                                         # misaligned views: element offset 1
                        sb, yb = torch.empty(n + 1, dtype=dtype, device=dev), torch.empty(n + 1, dtype=dtype, device=dev)
                        sb[1:].copy_(T(s, dev)); yb[1:].copy_(T(y, dev))
                        lo.push(ops[fused], sb[1:], yb[1:])
                    else:
                        lo.push(ops[fused], T(s, dev), T(y, dev))
                    assert ops[fused]._last_push_accepted == want, (n, k, fused, want)
                    assert ops[fused].data.insert == Oo.insert, (n, k, fused)
                if k in (0, 1, 3, 5, mem - 1, mem, len(prs) - 1):
                    got = {}
                    for fused in (1, 0):
                        res = torch.full((n,), float("nan"), dtype=dtype, device=dev)
                        lo.mul(res, ops[fused], T(x, dev), 1.0, 0.0)
                        got[fused] = res.cpu().numpy()
                    ref = Oo.mul(np.empty(n, dtype=npd), x, 1.0, 0.0)
                    assert rel(got[1], ref) <= tol and rel(got[0], ref) <= tol, (n, k, rel(got[1], ref), rel(got[0], ref))
                    assert rel(got[1], got[0]) <= same, (n, k, rel(got[1], got[0]))
            assert nacc >= mem + 1
            for fused in (1, 0):
                assert abs(ops[fused].data.scaling_factor - Oo.scaling_factor) <= 1e-6 * abs(Oo.scaling_factor)
            d1, d0 = lo.diag(ops[1]).cpu().numpy(), lo.diag(ops[0]).cpu().numpy()
            assert rel(d1, Oo.diag()) <= tol and rel(d1, d0) <= same
            # the panels hold exactly the accepted pairs (the inserts rode in the rebuild) and the padding stayed zero
            for slot in range(mem):
                for which in ("s", "y"):
                    assert torch.equal(ops[1].data.column(which, slot), ops[0].data.column(which, slot)), (n, which, slot)
            lo.reset(ops[1]); Oo.reset()
            for s, y in prs[:2]:
                lo.push(ops[1], T(s, dev), T(y, dev)); Oo.push(s, y)
            res = torch.empty(n, dtype=dtype, device=dev)
            lo.mul(res, ops[1], T(x, dev), 1.0, 0.0)
            assert rel(res.cpu().numpy(), Oo.mul(np.empty(n, dtype=npd), x, 1.0, 0.0)) <= tol
    finally:
        ctx.tune("push_fused", 1)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_rejected_push_leaves_stale_gram_state_alone(lo, dev, dtype):
    """Regression (found by MXLO_QNFUZZ32_SEEDS=4010, seed 1680): after a reference-ordered push! the Gram matrices and
    the a_k coefficients are stale (`gram_ok` false) and solve_shifted_system! rebuilds BOTH on its next call. The first
    one-pass push! made the Gram matrices consistent before its accept / reject decision — a REJECTED pair then left
    `gram_ok` true with stale coefficients and the next solve was wrong by tens of percent. A rejected push! must not
    touch any state: such pushes now take the two-kernel schedule, which rebuilds only for an accepted pair."""
    npd = NP[dtype]
    n, mem = 203, 6
    rng = np.random.default_rng(1680)
    B = lo.LBFGSOperator(dtype, n, mem=mem, scaling=True, device=dev)
    Bo = oracle.LBFGS(n, mem=mem, scaling=True, inverse=False, dtype=npd)
    prs = pairs(rng, n, 6, npd)
    for k, (s, y) in enumerate(prs):
        B.set_push_mode("reforder" if k in (3, 4) else "compact")
        lo.push(B, T(s, dev), T(y, dev)); Bo.push(s, y)
    B.set_push_mode("compact")
    lo.push(B, T(prs[0][0], dev), T(-prs[0][0], dev))                      # y's < 0: rejected, nothing may change
    assert not Bo.push(prs[0][0], (-prs[0][0]).astype(npd)) or True
    assert B.data.insert == Bo.insert
    x = rng.uniform(-1, 1, n).astype(npd)
    b = Bo.mul(np.empty(n, npd), x) + npd(0.5) * x
    got = lo.solve_shifted_system(torch.zeros(n, dtype=dtype, device=dev), B, T(b.astype(npd), dev), npd(0.5)).cpu().numpy()
    want = Bo.solve_shifted(np.zeros(n, npd), b.astype(npd), npd(0.5))
    assert rel(got, want) <= (1e-8 if dtype == torch.float64 else 1e-3)
    res = torch.empty(n, dtype=dtype, device=dev)
    lo.mul(res, B, T(x, dev), 1.0, 0.0)
    assert rel(res.cpu().numpy(), Bo.mul(np.empty(n, npd), x)) <= (1e-9 if dtype == torch.float64 else QN_F32)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("kind", ["inv", "fwd", "lsr1"])
@pytest.mark.parametrize("n,mem", [(1, 1), (7, 3), (4096, 5), (65_536, 5), (100_003, 20), (131_071, 32), (140_001, 5),
                                   (9_001, 10), (20_000, 20), (30_011, 16), (1_000, 6), (5_000, 12), (8_000, 28)])     # round 4: the (1, 20) and (2, 12) batch variants
def test_single_launch_dots_and_coefficients_match_the_four_launch_apply(lo, dev, dtype, kind, n, mem):
    """for launch-bound sizes (<= 64 workgroups, <= 40 panel columns) a quasi-Newton apply is ONE launch
    (csrc/qn.hip: qn_apply_fused_kernel — partial dots, fence-free slot exchange, fixed-order finalize, the coefficient
    recurrence on one wave per workgroup, the combine on the workgroup's slice) instead of 4. Against the four-launch schedule
    (`qn_fused_small` = 0: same coefficient code, different summation order of the dots) and the oracle; partially filled
    and wrapped memories, alpha/beta forms, the fused shifted apply, a misaligned x (falls back), sizes on both sides of
    the 64-workgroup limit, and bit-identical results from run to run."""
    from linearoperators_jl_amd.device import get_ctx
    ctx = get_ctx(dev)
    npd = NP[dtype]
    if kind == "lsr1" and n <= mem:
        pytest.skip("SR1 with more pairs than dimensions is rounding noise in the reference too")
    rng = np.random.default_rng(n + 31 * mem)
    make = {"inv": lo.InverseLBFGSOperator, "fwd": lo.LBFGSOperator, "lsr1": lo.LSR1Operator}[kind]
    op = make(dtype, n, mem=mem, scaling=True, device=dev)
    O = oracle.LSR1(n, mem=mem, scaling=True, dtype=npd) if kind == "lsr1" else oracle.LBFGS(n, mem=mem, scaling=True, inverse=(kind == "inv"), dtype=npd)
    tol = 1e-9 if dtype == torch.float64 else QN_F32
    x, r0 = rng.uniform(-1, 1, n).astype(npd), rng.uniform(-1, 1, n).astype(npd)
    xm = torch.empty(n + 1, dtype=dtype, device=dev)
    xm[1:].copy_(T(x, dev))
    for k, (s, y) in enumerate(pairs(rng, n, mem + 2, npd)):
        lo.push(op, T(s, dev), T(y, dev)); O.push(s, y)
        if k not in (0, mem // 2, mem + 1):
            continue
        for a, b in ((1.0, 0.0), (2.0, -3.0)):
            fl = oracle.SCALARS_F64 if dtype == torch.float32 else 0
            want = O.mul(r0.copy(), x, a, b, flags=fl)
            got = {}
            for fused in (1, 0):
                ctx.tune("qn_fused_small", fused)
                try:
                    res = T(r0.copy(), dev)
                    lo.mul(res, op, T(x, dev), a, b)
                    got[fused] = res.cpu().numpy()
                    if fused:
                        res2 = T(r0.copy(), dev)
                        lo.mul(res2, op, T(x, dev), a, b)
                        assert np.array_equal(res2.cpu().numpy(), got[1]), "run-to-run determinism"
                        res3 = T(r0.copy(), dev)
                        lo.mul(res3, op, xm[1:], a, b)                       # misaligned x: the four-launch path
                        assert rel(res3.cpu().numpy(), want) <= tol
                finally:
                    ctx.tune("qn_fused_small", 1)
            assert rel(got[1], want) <= tol and rel(got[0], want) <= tol, (k, a, b, rel(got[1], want), rel(got[0], want))
            assert rel(got[1], got[0]) <= (1e-12 if dtype == torch.float64 else 2e-5), (k, a, b)
    sh = lo.ShiftedOperator(op, 0.37)
    res = T(r0.copy(), dev)
    lo.mul(res, sh, T(x, dev), 1.5, 0.5)
    want = 1.5 * (O.mul(np.empty(n, npd), x).astype(np.float64) + 0.37 * x.astype(np.float64)) + 0.5 * r0.astype(np.float64)
    assert rel(res.cpu().numpy().astype(np.float64), want) <= tol


def test_single_launch_apply_exchange_stress(lo, dev):
    """The slot exchange of the single-launch quasi-Newton apply under churn: 24,000 back-to-back applies alternating
    between six operators (three kinds, different column counts, grid sizes from 1 to 49 workgroups, Float64 and
    Float32) that all share the ctx's two slot sets and its epoch word, interleaved with single-launch Householder
    applies (their own slots), four-launch applies (`qn_fused_small` toggled) and a long streaming kernel. It must not
    hang, and every operator's result must stay bit-identical to its first one (a stale or foreign partial would
    change the dots)."""
    from linearoperators_jl_amd.device import get_ctx
    ctx = get_ctx(dev)
    rng = np.random.default_rng(77)
    ops = []
    # (round 5: two operators large enough for the PERSISTENT launch, which shares the slot sets and the epoch word)
    for kind, dtype, n, mem in (("inv", torch.float64, 300, 3), ("fwd", torch.float64, 100_003, 20), ("lsr1", torch.float64, 4096, 7),
                                ("fwd", torch.float32, 65_536, 5), ("inv", torch.float32, 9_001, 10), ("lsr1", torch.float32, 70_001, 2),
                                ("inv", torch.float64, (1 << 19) + 5, 5), ("fwd", torch.float32, (1 << 20) + 3, 10)):
        npd = NP[dtype]
        make = {"inv": lo.InverseLBFGSOperator, "fwd": lo.LBFGSOperator, "lsr1": lo.LSR1Operator}[kind]
        op = make(dtype, n, mem=mem, device=dev)
        for s, y in pairs(rng, n, mem + 1, npd):
            lo.push(op, T(s, dev), T(y, dev))
        x = T(rng.uniform(-1, 1, n).astype(npd), dev)
        res = torch.empty(n, dtype=dtype, device=dev)
        lo.mul(res, op, x, 1.0, 0.0)
        ops.append((op, x, res, res.clone()))
    nh = 50_000
    h = rng.standard_normal(nh)
    H = lo.opHouseholder(T(h / np.linalg.norm(h), dev))
    hv, hr = T(rng.uniform(-1, 1, nh), dev), torch.empty(nh, dtype=torch.float64, device=dev)
    big = torch.rand(30_000_000, dtype=torch.float64, device=dev)
    D = lo.opDiagonal(big)
    bigr = torch.empty_like(big)
    torch.cuda.synchronize()
    for it in range(4000):
        for k, (op, x, res, _) in enumerate(ops):
            lo.mul(res, op, x, 1.0, 0.0)
            if (it + k) % 7 == 0:
                lo.mul(hr, H, hv, 1.0, 0.0)
        if it % 500 == 250:
            lo.mul(bigr, D, big, 1.0, 0.0)                       # a long kernel in front of the next fused launches
        if it % 1000 == 999:
            torch.cuda.synchronize()
            for op, x, res, first in ops:
                assert torch.equal(res, first), it
    ctx.tune("qn_fused_small", 0)
    ctx.tune("qn_persist", 0)
    try:
        for op, x, res, first in ops:
            lo.mul(res, op, x, 1.0, 0.0)
            tol = 1e-12 if res.dtype == torch.float64 else 2e-5
            assert (torch.linalg.vector_norm((res - first).double()) / torch.linalg.vector_norm(first.double())).item() <= tol
    finally:
        ctx.tune("qn_fused_small", 1)
        ctx.tune("qn_persist", 1)


@pytest.mark.parametrize("kind", ["inv", "fwd", "lsr1"])
def test_single_launch_apply_timeout_is_an_error_not_a_hang(lo, dev, kind):
    """the single-launch quasi-Newton apply must END when a peer workgroup never publishes (test hook
    `fused_debug_drop`): NaN result, ctx fault word, an error naming the timeout at the next call, exchange state
    re-armed, the ctx usable afterwards (four launches), and the single launch usable again once re-enabled."""
    from linearoperators_jl_amd.device import get_ctx
    ctx = get_ctx(dev)
    rng = np.random.default_rng(5)
    n, mem = 20_000, 4                                   # 10 workgroups
    make = {"inv": lo.InverseLBFGSOperator, "fwd": lo.LBFGSOperator, "lsr1": lo.LSR1Operator}[kind]
    op = make(torch.float64, n, mem=mem, device=dev)
    for s, y in pairs(rng, n, mem + 1, np.float64):
        lo.push(op, T(s, dev), T(y, dev))
    x = T(rng.uniform(-1, 1, n), dev)
    res = torch.zeros(n, dtype=torch.float64, device=dev)
    try:
        lo.mul(res, op, x, 1.0, 0.0)
        torch.cuda.synchronize()
        good = res.clone()
        ctx.tune("fused_timeout_ms", 20)
        ctx.tune("fused_debug_drop", 3)
        lo.mul(res, op, x, 1.0, 0.0)
        torch.cuda.synchronize()
        assert bool(torch.isnan(res).any())
        ctx.tune("fused_debug_drop", -1)
        with pytest.raises(Exception, match="timed out"):
            lo.mul(res, op, x, 1.0, 0.0)
        lo.mul(res, op, x, 1.0, 0.0)                     # four launches now
        torch.cuda.synchronize()
        assert float((res - good).norm() / good.norm()) <= 1e-13
        ctx.tune("qn_fused_small", 1)
        for _ in range(4):
            lo.mul(res, op, x, 1.0, 0.0)
        torch.cuda.synchronize()
        assert torch.equal(res, good)                    # the re-armed single launch is bit-identical to the first one
    finally:
        ctx.tune("fused_debug_drop", -1)
        ctx.tune("fused_timeout_ms", 2000)
        for key in ("house_fused", "qn_fused_small", "qn_persist", "herm_single", "kron_fuse"):   # a fault switches them all off
            ctx.tune(key, 1)


# ------------------------------------------------------------------------------- the persistent single-launch apply
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("kind", ["inv", "fwd", "lsr1"])
@pytest.mark.parametrize("n,mem", [(1 << 19, 5), ((1 << 19) + 5, 10), ((1 << 20) + 1, 3), (3 * (1 << 18) + 7, 20), (1_000_003, 7)])
def test_persistent_apply_matches_the_four_launch_apply_and_the_oracle(lo, dev, dtype, kind, n, mem):
    """at cache-resident sizes (n >= 2^19, panel <= 448 MB) a quasi-Newton apply is ONE persistent launch
    (csrc/qn.hip: qn_apply_persist_kernel — one 512-thread workgroup per CU owning a contiguous run of chunks; dots,
    the slot exchange of the single-launch apply, the coefficient recurrence per workgroup, the combine back to front).
    Against the four-launch schedule (`qn_persist` = 0: same coefficient code and elementwise formulas, a different fixed
    summation order of the dots) and the oracle (src/lbfgs.jl:117-154,173-202; src/lsr1.jl:89-107): partially filled and
    wrapped memories, alpha / beta forms, the fused shifted apply, lengths that end in a partial vector and a partial
    chunk, a misaligned x (four launches), and bit-identical results run to run."""
    from linearoperators_jl_amd.device import get_ctx
    ctx = get_ctx(dev)
    npd = NP[dtype]
    rng = np.random.default_rng(n + 31 * mem)
    make = {"inv": lo.InverseLBFGSOperator, "fwd": lo.LBFGSOperator, "lsr1": lo.LSR1Operator}[kind]
    op = make(dtype, n, mem=mem, scaling=True, device=dev)
    O = oracle.LSR1(n, mem=mem, scaling=True, dtype=npd) if kind == "lsr1" else oracle.LBFGS(n, mem=mem, scaling=True, inverse=(kind == "inv"), dtype=npd)
    tol = 1e-9 if dtype == torch.float64 else QN_F32
    x, r0 = rng.uniform(-1, 1, n).astype(npd), rng.uniform(-1, 1, n).astype(npd)
    xm = torch.empty(n + 1, dtype=dtype, device=dev)
    xm[1:].copy_(T(x, dev))
    def launches():
        import ctypes as C
        a = (C.c_int64 * 12)()
        lo._lib.call("mxlo_debug_counters", a)
        return a[10]
    ctx.tune("qn_persist_min_bytes", 0)          # (default 32 MiB: a two-column panel would take the other forms) — restored below
    for k, (s, y) in enumerate(pairs(rng, n, mem + 2, npd)):
        lo.push(op, T(s, dev), T(y, dev)); O.push(s, y)
        if k not in (0, mem // 2, mem + 1):
            continue
        for a, b in ((1.0, 0.0), (2.0, -3.0)):
            fl = oracle.SCALARS_F64 if dtype == torch.float32 else 0
            want = O.mul(r0.copy(), x, a, b, flags=fl)
            got = {}
            for persist in (1, 0):
                ctx.tune("qn_persist", persist)
                ctx.tune("qn_fused_small", persist)      # (n = 2^19 is still within reach of the single-launch slice form)
                try:
                    res = T(r0.copy(), dev)
                    xd = T(x, dev)
                    l0 = launches()
                    lo.mul(res, op, xd, a, b)
                    nl = launches() - l0
                    assert (nl == 1) if persist else (nl >= 3), (persist, nl)
                    got[persist] = res.cpu().numpy()
                    if persist:
                        res2 = T(r0.copy(), dev)
                        lo.mul(res2, op, xd, a, b)
                        assert np.array_equal(res2.cpu().numpy(), got[1]), "run-to-run determinism"
                        res3 = T(r0.copy(), dev)
                        lo.mul(res3, op, xm[1:], a, b)                       # misaligned x: the four-launch path
                        assert rel(res3.cpu().numpy(), want) <= tol
                finally:
                    ctx.tune("qn_persist", 1)
                    ctx.tune("qn_fused_small", 1)
            assert rel(got[1], want) <= tol and rel(got[0], want) <= tol, (k, a, b, rel(got[1], want), rel(got[0], want))
            assert rel(got[1], got[0]) <= (1e-12 if dtype == torch.float64 else 2e-5), (k, a, b)
    sh = lo.ShiftedOperator(op, 0.37)
    res = T(r0.copy(), dev)
    lo.mul(res, sh, T(x, dev), 1.5, 0.5)
    want = 1.5 * (O.mul(np.empty(n, npd), x).astype(np.float64) + 0.37 * x.astype(np.float64)) + 0.5 * r0.astype(np.float64)
    assert rel(res.cpu().numpy().astype(np.float64), want) <= tol
    ctx.tune("qn_persist_min_bytes", 32 << 20)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("kind", ["inv", "fwd", "lsr1"])
@pytest.mark.parametrize("n,mem", [(1 << 19, 2), ((1 << 19) + 5, 10), (700_001, 3), ((1 << 20) + 1, 5), (1_500_003, 7), ((1 << 21) + 2, 5),
                                   ((1 << 22) + 9, 2)])
def test_persistent_apply_with_lds_parking_has_the_bits_of_the_plain_form(lo, dev, dtype, kind, n, mem):
    """the dots phase of the persistent apply parks x and the first columns of the combine order in the CU's
    LDS (18 tiles of 8 KiB; csrc/qn.hip, `lds_k`) and the combine phase reads them from there. Values and order of every
    operation are those of `qn_persist_lds` = 0: bit-identical — for every count of parked columns the kernel has a case
    for (chunks per workgroup 1 ... 16), lengths that end in a partial chunk (never parked), alpha / beta forms and the
    fused shifted apply, and with memories that are partially filled."""
    from linearoperators_jl_amd.device import get_ctx
    ctx = get_ctx(dev)
    npd = NP[dtype]
    rng = np.random.default_rng(n + 17 * mem)
    make = {"inv": lo.InverseLBFGSOperator, "fwd": lo.LBFGSOperator, "lsr1": lo.LSR1Operator}[kind]
    op = make(dtype, n, mem=mem, scaling=True, device=dev)
    x, r0 = T(rng.uniform(-1, 1, n).astype(npd), dev), T(rng.uniform(-1, 1, n).astype(npd), dev)
    ctx.tune("qn_persist_min_bytes", 0)
    try:
        for k, (s, y) in enumerate(pairs(rng, n, mem + 1, npd)):
            lo.push(op, T(s, dev), T(y, dev))
            if k not in (0, mem):
                continue
            for target, a, b in ((op, 1.0, 0.0), (op, 0.7, -1.3), (lo.ShiftedOperator(op, 0.37), 1.5, 0.5)):
                got = {}
                for park in (1, 0):
                    ctx.tune("qn_persist_lds", park)
                    res = r0.clone()
                    lo.mul(res, target, x, a, b)
                    got[park] = res
                assert torch.equal(got[1], got[0]), (k, a, b)
    finally:
        ctx.tune("qn_persist_lds", 1)
        ctx.tune("qn_persist_min_bytes", 32 << 20)


def test_persistent_apply_timeout_is_an_error_not_a_hang(lo, dev):
    """The persistent apply shares the bounded wait of the single-launch forms: a workgroup that never publishes (test
    hook `fused_debug_drop`) ends the launch with NaN + the ctx fault word; the next call reports it, re-arms the slots
    and switches the single-launch forms off; four launches give the right answer; re-enabled, the persistent launch is
    bit-identical to its first run. A captured graph of it replays (epoch word in device memory)."""
    from linearoperators_jl_amd.device import get_ctx
    ctx = get_ctx(dev)
    rng = np.random.default_rng(6)
    n, mem = (1 << 19) + 3, 4
    op = lo.LBFGSOperator(torch.float64, n, mem=mem, device=dev)
    for s, y in pairs(rng, n, mem + 1, np.float64):
        lo.push(op, T(s, dev), T(y, dev))
    x = T(rng.uniform(-1, 1, n), dev)
    res = torch.zeros(n, dtype=torch.float64, device=dev)
    try:
        lo.mul(res, op, x, 1.0, 0.0)
        torch.cuda.synchronize()
        good = res.clone()
        ctx.tune("fused_timeout_ms", 20)
        ctx.tune("fused_debug_drop", 100)
        lo.mul(res, op, x, 1.0, 0.0)
        torch.cuda.synchronize()
        assert bool(torch.isnan(res).any())
        ctx.tune("fused_debug_drop", -1)
        with pytest.raises(Exception, match="timed out"):
            lo.mul(res, op, x, 1.0, 0.0)
        lo.mul(res, op, x, 1.0, 0.0)                     # four launches now
        torch.cuda.synchronize()
        assert float((res - good).norm() / good.norm()) <= 1e-12
        ctx.tune("qn_persist", 1)
        for _ in range(4):
            lo.mul(res, op, x, 1.0, 0.0)
        torch.cuda.synchronize()
        assert torch.equal(res, good)
    finally:
        ctx.tune("fused_debug_drop", -1)
        ctx.tune("fused_timeout_ms", 2000)
        ctx.tune("house_fused", 1)
        ctx.tune("qn_fused_small", 1)
        ctx.tune("qn_persist", 1)
        ctx.tune("herm_single", 1)
        ctx.tune("kron_fuse", 1)


@pytest.mark.parametrize("kind", ["inv", "fwd", "lsr1"])
def test_push_of_a_pair_that_lives_in_the_operators_own_storage(lo, dev, kind):
    """the streaming push! schedules read the caller's s and y while kernels store into the slot being
    replaced (and, L-SR1, while the rebuild rewrites the a_k panel). A pair that is a VIEW of the operator's own panels
    (`mxlo_qn_column`) must therefore take the copy-based schedule — detected on the host by address overlap. Pushing
    column views, including the very slot that is about to be overwritten, must give what pushing private copies of
    the same data gives."""
    rng = np.random.default_rng(31)
    n, mem = 6_004, 3
    make = {"inv": lo.InverseLBFGSOperator, "fwd": lo.LBFGSOperator, "lsr1": lo.LSR1Operator}[kind]
    A, B = make(torch.float64, n, mem=mem, device=dev), make(torch.float64, n, mem=mem, device=dev)
    for s, y in pairs(rng, n, mem, np.float64):
        lo.push(A, T(s, dev), T(y, dev))
        lo.push(B, T(s, dev), T(y, dev))
    x = T(rng.uniform(-1, 1, n), dev)
    for step in range(2 * mem):
        ins = A.data.insert - 1                               # 0-based slot the next push overwrites
        k = (ins + step) % mem                                # step % mem == 0: the slot that is being replaced
        sv, yv = A.data.column("s", k), A.data.column("y", k)
        # a new pair built from stored columns: s from the panel (a view), y = stored y + a little of s (private copy)
        ynew = (B.data.column("y", k) + 0.125 * B.data.column("s", k)).clone()
        ypriv = ynew.clone()
        spriv = B.data.column("s", k).clone()
        lo.push(A, sv, ynew)                                  # s aliases A's own S panel
        lo.push(B, spriv, ypriv)
        assert A._last_push_accepted == B._last_push_accepted, step
        ra, rb = torch.empty_like(x), torch.empty_like(x)
        lo.mul(ra, A, x, 1.0, 0.0)
        lo.mul(rb, B, x, 1.0, 0.0)
        assert float((ra - rb).norm() / rb.norm()) <= 1e-11, (kind, step)
        for which in ("s", "y"):
            for slot in range(mem):
                assert torch.equal(A.data.column(which, slot), B.data.column(which, slot)), (kind, step, which, slot)
        del yv


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("scaling", [True, False])
def test_lsr1_streaming_push_decisions_agree_with_the_apply_based_schedule(lo, dev, dtype, scaling):
    """the streaming L-SR1 push forms r = y - B s with a_k's from the Gram data; its rounding differs from
    the apply-based evaluation, and r's = y's - s'B s is a cancellation whose absolute error is ~ eps (|y| + |B s|) |s|
    while the thresholds of src/lsr1.jl:131-141 are ~ eps: for a pair the memory (nearly) reproduces the decision is made
    by rounding noise. The library therefore re-evaluates every push that does not clear its thresholds by 2^8 error
    budgets with the apply-based schedule (`lsr1_decision_is_marginal`). Checked here on sequences rich in borderline
    pushes (the same pair again, y = B s + 1e-15 … 1e-6, y nearly orthogonal / parallel to s): whenever a push is decided
    — by a margin computed independently in the test — `push_fused` 1 and 0 take the same decision and keep the same
    insert position; inside the noise band either decision is legitimate (the reference's own depends on its BLAS's
    summation order), so there the two operators are only required to stay usable, and are reset if they part ways."""
    from linearoperators_jl_amd.device import get_ctx
    ctx = get_ctx(dev)
    npd = NP[dtype]
    eps = float(np.finfo(npd).eps)
    same = 1e-9 if dtype == torch.float64 else 2e-3
    decided = gray = 0
    try:
        for seed in range(12):
            rng = np.random.default_rng(900 + seed)
            n, mem = int(rng.integers(200, 5000)) * 2, int(rng.integers(2, 8))
            ops = {}
            for fused in (1, 0):
                ctx.tune("push_fused", fused)
                ops[fused] = lo.LSR1Operator(dtype, n, mem=mem, scaling=scaling, device=dev)
            hist = []
            tainted = False                                    # a pair decided by noise sits in one of the memories
            for k in range(3 * mem + 6):
                c = int(rng.integers(0, 7))
                s = rng.uniform(-1, 1, n).astype(npd)
                if c <= 1 or not hist:
                    y = ((0.5 + 1.5 * rng.random(n)) * s + 1e-2 * rng.standard_normal(n)).astype(npd)
                elif c == 2:                                   # the same pair again: r is rounding noise
                    s, y = hist[int(rng.integers(len(hist)))]
                elif c == 3:                                   # y = B s (+ a perturbation from 1e-15 to 1e-6): r tiny
                    Bs = torch.empty(n, dtype=dtype, device=dev)
                    lo.mul(Bs, ops[0], T(s, dev), 1.0, 0.0)
                    y = (Bs.cpu().numpy() + npd(10.0 ** rng.uniform(-15, -6)) * rng.standard_normal(n).astype(npd)).astype(npd)
                elif c == 4:                                   # y almost orthogonal to s
                    y = rng.standard_normal(n).astype(npd)
                    y = (y - (y @ s) / (s @ s) * s * npd(1.0 - 10.0 ** rng.uniform(-16, -3))).astype(npd)
                elif c == 5:                                   # y almost parallel to s
                    y = (npd(2.5) * s + npd(10.0 ** rng.uniform(-16, -4)) * rng.standard_normal(n).astype(npd)).astype(npd)
                else:                                          # y orthogonal to s up to a few eps: y's ≈ eps |y||s| (:137)
                    y = rng.standard_normal(n).astype(npd)
                    y = (y - (y @ s) / (s @ s) * s).astype(npd)
                    y = (y + npd(eps * 10.0 ** rng.uniform(-1, 2)) * np.sqrt(y @ y / (s @ s)).astype(npd) * s).astype(npd)
                # independent margin (float64 on the host, r through the apply-based operator)
                Bs = torch.empty(n, dtype=dtype, device=dev)
                lo.mul(Bs, ops[0], T(s, dev), 1.0, 0.0)
                s64, y64 = s.astype(np.float64), y.astype(np.float64)
                with np.errstate(all="ignore"):
                    r64 = y64 - Bs.cpu().numpy().astype(np.float64)
                sN, yN, rN = np.linalg.norm(s64), np.linalg.norm(y64), np.linalg.norm(r64)
                budget = 256 * eps * ((2 * yN + rN) * sN + 1)
                clear = np.isfinite(rN) and abs(r64 @ s64) > 16 * ((eps + eps * rN * sN) + budget)
                if scaling and clear:
                    ys, yy = y64 @ s64, y64 @ y64
                    thr = eps * yN * sN
                    clear = abs(ys) > 16 * thr * 257 and \
                        np.linalg.norm(y64 - s64 * (yy / ys)) > 16 * (thr + 256 * eps * (yN + sN * yy / abs(ys)))
                acc = {}
                for fused in (1, 0):
                    ctx.tune("push_fused", fused)
                    lo.push(ops[fused], T(s, dev), T(y, dev))
                    acc[fused] = ops[fused]._last_push_accepted
                if clear:
                    decided += 1
                    assert acc[1] == acc[0], (seed, k, c, acc)
                    assert ops[1].data.insert == ops[0].data.insert
                else:
                    gray += 1
                    if acc[1] != acc[0]:                       # noise decided differently: legitimate, start over
                        for fused in (1, 0):
                            lo.reset(ops[fused])
                        hist = []
                        tainted = False
                        continue
                    tainted = tainted or acc[1]
                if acc[1]:
                    hist.append((s, y))
            x = T(rng.uniform(-1, 1, n).astype(npd), dev)
            r1, r0 = torch.empty_like(x), torch.empty_like(x)
            lo.mul(r1, ops[1], x, 1.0, 0.0)
            lo.mul(r0, ops[0], x, 1.0, 0.0)
            if not tainted:                                    # every pair in the memories was decided by a clear margin
                assert bool(torch.isfinite(r0).all()) and bool(torch.isfinite(r1).all()), seed
                assert float((r1 - r0).norm()) <= same * max(float(r0.norm()), float(x.norm())), seed
        assert decided >= 40 and gray >= 20, (decided, gray)
    finally:
        ctx.tune("push_fused", 1)
