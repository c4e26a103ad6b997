"""-m gpu: diagonal quasi-Newton operators (src/DiagonalHessianApproximation.jl) through the C ABI.

mul! is mxlo_diag_mul (bit-exact vs the oracle); push! is one fused reduction pass + the reference's scalar
recurrence + one update pass (mxlo_diagqn_push): the reductions are fixed-order sums, so the updated diagonal is
compared to the oracle's statement-by-statement NumPy restatement at 1e-12 (fp64) / 2e-5 (fp32) relative, to the
reference test's hard-coded answers (test_diag.jl:75-106) at its own 1e-10, and through the weak secant equation."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
NP = {torch.float64: np.float64, torch.float32: np.float32}


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    nb = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (nb if nb else 1.0)


def test_kat_hard_coded(lo, dev, kat):
    """test_diag.jl:75-106 and the weak secant sets :52-72."""
    for c in [c for c in kat if c["kind"] == "diagqn_push"]:
        s, y = T(np.array(c["s"]), dev), T(np.array(c["y"]), dev)
        for ctor, key in ((lo.DiagonalPSB, "expect_psb"), (lo.DiagonalAndrei, "expect_andrei")):
            B = ctor(T(np.array(c["d0"]), dev))
            assert lo.push(B, s, y) is B
            assert np.linalg.norm(B.d.cpu().numpy() - np.array(c[key])) <= 1e-10
            assert abs(float(torch.dot(s, B * s)) - float(torch.dot(s, y))) <= 1e-10
            assert lo.isallocated5(B) and lo.has_args5(B) and lo.issymmetric(B) and lo.ishermitian(B)
        S = lo.SpectralGradient(1.0, 3, device=dev)
        lo.push(S, s, y)
        assert abs(float(S.d[0]) - c["expect_spectral"]) <= 1e-10
        assert rel((S * s).cpu().numpy(), c["expect_spectral"] * np.array(c["s"])) <= 1e-15


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("n", [1, 5, 1000, 4099, 1_048_577])
@pytest.mark.parametrize("kind", ["psb", "andrei", "bfgs", "spectral"])
def test_push_and_mul_vs_oracle(lo, dev, dtype, n, kind):
    npd = NP[dtype]
    rng = np.random.default_rng(n)
    s = rng.uniform(-1, 1, n).astype(npd)
    y = (s * rng.uniform(0.5, 2.0, n)).astype(npd)            # s'y > 0
    d0 = rng.uniform(0.5, 1.5, n).astype(npd)
    tol = 1e-12 if dtype == torch.float64 else 2e-5
    if kind == "spectral":
        B = lo.SpectralGradient(npd(1.5), n, dtype=dtype, device=dev)
        O = oracle.DiagonalQN(kind, np.array([1.5], dtype=npd))
    else:
        ctor = {"psb": lo.DiagonalPSB, "andrei": lo.DiagonalAndrei, "bfgs": lo.DiagonalBFGS}[kind]
        dd = T(d0, dev)
        B = ctor(dd)
        assert B.d.data_ptr() == dd.data_ptr()                # `d` is the operator's diagonal itself
        O = oracle.DiagonalQN(kind, d0)
    for it in range(3):                                        # repeated updates on the evolving diagonal
        lo.push(B, T(s, dev), T(y, dev))
        O.push(s, y)
        assert rel(B.d.cpu().numpy(), O.d) <= tol * (1 + it), (kind, it)
        s, y = y.copy(), (y * rng.uniform(0.5, 2.0, n).astype(npd)).astype(npd)
    # mul! on the updated diagonal: bit-exact against the oracle evaluated on the DEVICE's diagonal
    v, r0 = rng.uniform(-1, 1, n).astype(npd), rng.uniform(-1, 1, n).astype(npd)
    Od = oracle.DiagonalQN(kind, B.d.cpu().numpy())
    fl = oracle.SCALARS_F64 if dtype == torch.float32 else 0
    res = T(r0.copy(), dev)
    lo.mul(res, B, T(v, dev), 2.0, -3.0)
    assert np.array_equal(res.cpu().numpy(), Od.mul(r0.copy(), v, 2.0, -3.0, flags=fl))
    for op in (B, B.T, B.H):
        assert np.array_equal((op * T(v, dev)).cpu().numpy(), Od.mul(np.empty(n, npd), v))
    assert lo.nprod(B) == 4
    lo.reset(B)                                               # d .= 1 and the counters (:71-77)
    assert lo.nprod(B) == 0 and torch.equal(B.d, torch.ones_like(B.d))


def test_zero_step_raises_and_leaves_d(lo, dev):
    d = torch.full((64,), 2.0, dtype=torch.float64, device=dev)
    z = torch.zeros(64, dtype=torch.float64, device=dev)
    y = torch.ones(64, dtype=torch.float64, device=dev)
    for ctor in (lo.DiagonalPSB, lo.DiagonalAndrei, lo.DiagonalBFGS):
        B = ctor(d)
        with pytest.raises(RuntimeError, match="s=0"):
            lo.push(B, z, y)
        assert torch.equal(d, torch.full_like(d, 2.0))
    S = lo.SpectralGradient(3.0, 64, device=dev)
    with pytest.raises(RuntimeError, match="divide by zero"):
        lo.push(S, z, y)
    assert float(S.d[0]) == 3.0
    with pytest.raises(AssertionError):
        lo.SpectralGradient(-1.0, 4, device=dev)


def test_unaligned_views_and_big(lo, dev):
    """Views with different 16-byte phases (scalar path) and an HBM-sized push (nontemporal path)."""
    rng = np.random.default_rng(0)
    n = 10_001
    buf = [T(rng.uniform(0.5, 1.5, n + 3), dev) for _ in range(3)]
    d, s, y = buf[0][1:n + 1], buf[1][2:n + 2], buf[2][3:n + 3]
    O = oracle.DiagonalQN("psb", d.cpu().numpy()).push(s.cpu().numpy(), y.cpu().numpy())
    lo.push(lo.DiagonalPSB(d), s, y)
    assert rel(d.cpu().numpy(), O.d) <= 1e-12
    n = 20_000_003
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    s = torch.rand(n, dtype=torch.float64, device=dev, generator=g) - 0.5
    y = s * (torch.rand(n, dtype=torch.float64, device=dev, generator=g) + 0.5)
    B = lo.DiagonalAndrei(torch.ones(n, dtype=torch.float64, device=dev))
    lo.push(B, s, y)
    assert abs(float(torch.dot(s, B * s)) - float(torch.dot(s, y))) <= 1e-9 * float(torch.dot(s, y))   # weak secant
