"""Pins the CPU oracle (oracle/) against the known-answer cases held by the reference's own tests
(tests/golden/kat_reference_tests.json) and against independent dense NumPy constructions, the way
the reference's test-suite pins the package. CPU only."""
import numpy as np
import pytest

import oracle

SV = lambda n: np.array([-(-1.0) ** i for i in range(1, n + 1)])


def _by_kind(kat, kind):
    return [c for c in kat if c["kind"] == kind]


def test_kat_diag(kat):
    for c in _by_kind(kat, "diag"):
        d, u = np.array(c["d"]), np.array(c["u"])
        res = np.full_like(u, np.nan)                      # beta == 0 must not read res
        assert np.array_equal(oracle.diag_mul(res, d, u, 1.0, 0.0), np.array(c["expect_apply"])), c["name"]
        res = np.array(c["res0"])
        assert np.array_equal(oracle.diag_mul(res, d, u, c["alpha"], c["beta"]), np.array(c["expect_mul5"]))


def test_kat_diag_scalar(kat):
    for c in _by_kind(kat, "diag_scalar"):
        d, u = np.array(c["d"]), np.array(c["u"])
        res = np.empty_like(u)
        oracle.diag_mul(res, d, u, 1.0, 0.0, flags=oracle.D_SCALAR)
        assert np.array_equal(res, np.array(c["expect_apply"]))


def test_kat_diag_rect(kat):
    for c in _by_kind(kat, "diag_rect"):
        d, u, w = np.array(c["d"]), np.array(c["u"]), np.array(c["w"])
        nmin = min(c["nrow"], c["ncol"])
        res = np.full(c["nrow"], np.nan)
        oracle.diag_mul(res, d, u, 1.0, 0.0, n_min=nmin)
        assert np.array_equal(res, np.array(c["expect_apply"])), c["name"]
        res = np.full(c["ncol"], 7.0)                      # tail is zeroed even when beta != 0
        oracle.diag_mul(res, d, w, 1.0, 0.0, n_min=nmin)
        assert np.array_equal(res, np.array(c["expect_tapply"]))


def test_rect_diag_tail_zero_with_beta():
    res = np.full(5, 3.0)
    oracle.diag_mul(res, np.ones(3), np.ones(3), 1.0, 2.0, n_min=3)
    assert np.array_equal(res, [7.0, 7.0, 7.0, 0.0, 0.0])  # src/special-operators.jl:150


def test_kat_householder(kat):
    for c in _by_kind(kat, "householder"):
        h, u = np.array(c["h"]), np.array(c["u"])
        res = np.full_like(u, np.nan)
        assert np.array_equal(oracle.householder_mul(res, h, u, 1.0, 0.0), np.array(c["expect_apply"]))


def _idx_of(spec, n):
    if "list" in spec:
        return np.array(spec["list"], dtype=np.int64)
    if "range" in spec:
        a, b, s = spec["range"]
        return np.arange(a, b + 1, s, dtype=np.int64)
    if "scalar" in spec:
        return np.array([spec["scalar"]], dtype=np.int64)
    return np.arange(1, n + 1, dtype=np.int64)


def test_kat_restriction(kat):
    for c in _by_kind(kat, "restriction"):
        n = c["n"]
        idx = _idx_of(c["idx"], n)
        v = np.array(c["v"])
        w = oracle.restrict(np.empty(idx.size), v, idx)
        assert np.array_equal(w, np.array(c["expect_w"])), c["name"]                # P*v == v[idx]
        vz = oracle.extend(np.full(n, np.nan), w, idx)
        assert np.array_equal(vz, np.array(c["expect_vz"]))                           # P'*w == vz
        assert np.array_equal(oracle.restrict(np.empty(idx.size), vz, idx), w)        # (P*Z)*w == w


def test_extension_duplicates_last_write_wins():
    idx = np.array([2, 5, 2, 3], dtype=np.int64)
    u = np.array([10.0, 20.0, 30.0, 40.0])
    assert np.array_equal(oracle.extend(np.empty(6), u, idx), [0, 30.0, 40.0, 0, 20.0, 0])


def test_kat_lbfgs(kat):
    for c in _by_kind(kat, "lbfgs"):
        n, mem = c["n"], c["mem"]
        B = oracle.LBFGS(n, mem=mem, scaling=c["scaling"], inverse=False)
        H = oracle.LBFGS(n, mem=mem, scaling=c["scaling"], inverse=True)
        assert np.array_equal(B.dense(), np.eye(n)) and np.array_equal(H.dense(), np.eye(n))   # test_lbfgs.jl:18-19
        for p in c["pre_rejected"]:
            assert not B.push(np.array(p["s"]), np.array(p["y"])) and B.insert == 1            # :24-31
            assert not H.push(np.array(p["s"]), np.array(p["y"])) and H.insert == 1
        for p in c["pairs"]:
            assert B.push(np.array(p["s"]), np.array(p["y"]))
            assert H.push(np.array(p["s"]), np.array(p["y"]))
        assert B.insert == H.insert == c["expect_insert"]
        assert np.array_equal(B.ys, np.array(c["expect_ys_slots"]))
        v = np.array(c["v"])
        rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
        assert rel(B.mul(np.empty(n), v), np.array(c["expect_Bv"])) <= 1e-12
        assert rel(H.mul(np.empty(n), v), np.array(c["expect_Hv"])) <= 1e-12
        assert rel(B.diag(), np.array(c["expect_diagB"])) <= 1e-12
        assert np.linalg.norm(H.dense() @ B.dense() - np.eye(n)) <= np.sqrt(np.finfo(float).eps)  # :56
        assert np.linalg.norm(B.dense(), 2) <= B.opnorm_upper_bound                              # :70
        B.reset(); H.reset()
        assert B.scaling_factor == 1.0 and B.insert == 1
        assert np.linalg.norm(B.mul(np.empty(n), v) - v) < 1e-8 and np.linalg.norm(H.mul(np.empty(n), v) - v) < 1e-8
    for c in _by_kind(kat, "lbfgs_identity"):
        n = c["n"]
        LB = oracle.LBFGS(n, mem=c["mem"], scaling=False, inverse=False)
        for p in c["pairs"]:
            LB.push(np.array(p["s"]), np.array(p["y"]))
            assert np.linalg.norm(LB.dense() - np.eye(n)) < 1e-8 * np.sqrt(n)
        v = np.array(c["v"])
        assert np.allclose(LB.mul(np.empty(n), v), np.array(c["expect_Bv"]), rtol=1e-12)


def test_kat_solve_shifted_system(kat):
    """test/test_solve_shifted_system.jl:5-61 with deterministic pairs: the exact rational solution of (B + σI) x = b for
    the dense BFGS matrix of the kept pairs (scaling off and on), σ = 0 through ldiv! against the inverse operator."""
    cs = _by_kind(kat, "solve_shifted")
    assert len(cs) == 5
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    for c in cs:
        n = c["n"]
        B = oracle.LBFGS(n, mem=c["mem"], scaling=c["scaling"], inverse=False)
        H = oracle.LBFGS(n, mem=c["mem"], scaling=False, inverse=True)
        for p in c["pairs"]:
            assert B.push(np.array(p["s"]), np.array(p["y"]))
            H.push(np.array(p["s"]), np.array(p["y"]))
        b = np.array(c["b"])
        x = B.solve_shifted(np.zeros(n), b, c["sigma"])
        assert np.all(np.isfinite(x)) and rel(x, np.array(c["expect_x"])) <= 1e-10, c["name"]
        assert np.allclose(x, np.array(c["x_true"]), atol=1e-6, rtol=1e-6)                 # the reference's own assertion
        if "expect_Hb" in c:
            assert rel(H.mul(np.empty(n), b), np.array(c["expect_Hb"])) <= 1e-10
            assert np.allclose(x, H.mul(np.empty(n), b), atol=1e-6, rtol=1e-6)             # ldiv! test (:49-60)
    with pytest.raises(ValueError):
        B.solve_shifted(np.zeros(n), b, -0.1)                                              # ArgumentError (:43-47)


def test_kat_lsr1(kat):
    for c in _by_kind(kat, "lsr1"):
        n = c["n"]
        B = oracle.LSR1(n, mem=c["mem"], scaling=c["scaling"])
        s = SV(n)
        y = B.mul(np.empty(n), s)
        assert not B.push(s, y) and B.insert == 1                                   # test_lsr1.jl:18-21
        nacc = sum(B.push(np.array(p["s"]), np.array(p["y"])) for p in c["pairs"])
        assert nacc == c["expect_naccepted"] and B.insert == c["expect_insert"]
        v = np.array(c["v"])
        rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
        assert rel(B.mul(np.empty(n), v), np.array(c["expect_Bv"])) <= 1e-12
        assert rel(B.diag(), np.array(c["expect_diagB"])) <= 1e-12
        assert np.linalg.norm(B.dense(), 2) <= B.opnorm_upper_bound


# ------------------------------------------------------------------ dense-model pins (reference test style)
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_lbfgs_vs_dense_bfgs_random(dtype):
    """test_lbfgs.jl:73-99 with random well-conditioned pairs and scaling (gamma != 1)."""
    rng = np.random.default_rng(1)
    n, mem = 12, 12
    LB = oracle.LBFGS(n, mem=mem, scaling=False, inverse=False, dtype=dtype)
    LH = oracle.LBFGS(n, mem=mem, scaling=False, inverse=True, dtype=dtype)
    Bd = np.eye(n)
    for _ in range(mem):
        s = rng.uniform(-1, 1, n)
        y = s * rng.uniform(0.5, 2.0, n)
        Bs = Bd @ s
        Bd = Bd - np.outer(Bs, Bs) / (s @ Bs) + np.outer(y, y) / (y @ s)
        LB.push(s.astype(dtype), y.astype(dtype))
        LH.push(s.astype(dtype), y.astype(dtype))
    tol = 1e-10 if dtype == np.float64 else 2e-3
    assert np.linalg.norm(LB.dense() - Bd) < tol * np.linalg.norm(Bd)
    assert np.linalg.norm(LH.dense() @ Bd - np.eye(n)) < tol * 50


def test_lbfgs_scaling_and_wrap_inverse_consistency():
    rng = np.random.default_rng(2)
    n, mem = 30, 4
    B = oracle.LBFGS(n, mem=mem, scaling=True, inverse=False)
    H = oracle.LBFGS(n, mem=mem, scaling=True, inverse=True)
    for _ in range(mem + 3):
        s = rng.uniform(-1, 1, n)
        y = s * rng.uniform(0.5, 2.0, n) + 1e-2 * rng.standard_normal(n)
        assert B.push(s, y) and H.push(s, y)
    assert np.linalg.norm(H.dense() @ B.dense() - np.eye(n)) < 1e-9
    assert np.allclose(B.diag(), np.diag(B.dense()), rtol=1e-12)


def test_solve_shifted_roundtrip_and_ldiv():
    """test_solve_shifted_system.jl:5-61."""
    rng = np.random.default_rng(3)
    n, M = 100, 5
    for scaling in (False, True):
        B = oracle.LBFGS(n, mem=M, scaling=scaling, inverse=False)
        H = oracle.LBFGS(n, mem=M, scaling=scaling, inverse=True)
        for _ in range(10):
            s, y = rng.random(n), rng.random(n)
            B.push(s, y); H.push(s, y)
        x = rng.standard_normal(n)
        for sigma in (0.1, 0.0, 3.0):
            b = B.mul(np.empty(n), x) + sigma * x
            xs = B.solve_shifted(np.zeros(n), b, sigma)
            assert np.allclose(xs, x, atol=1e-6, rtol=1e-6)
        b = B.mul(np.empty(n), x)
        assert np.allclose(B.solve_shifted(np.zeros(n), b, 0.0), H.mul(np.empty(n), b), atol=1e-6, rtol=1e-6)
        with pytest.raises(ValueError):
            B.solve_shifted(np.zeros(n), b, -0.1)


def test_damped_lbfgs_properties():
    """test_lbfgs.jl:104-159: damped pairs keep B positive definite and H*B ≈ I."""
    n, mem = 10, 10
    B = oracle.LBFGS(n, mem=mem, damped=True, scaling=False, inverse=False, sigma2=0.8, sigma3=np.inf)
    H = oracle.LBFGS(n, mem=mem, damped=True, scaling=False, inverse=True, sigma2=0.8, sigma3=np.inf)
    rng = np.random.default_rng(4)
    for i in range(1, mem + 3):
        s = SV(n) * i
        y = rng.uniform(-1, 1, n)
        if s @ y <= 0:
            y = -y
        B.push(s.copy(), y.copy())
        g = -(B.dense() @ s)          # so that Bs = -alpha*g with alpha = 1 matches the forward Bs
        H.push(s.copy(), y.copy(), alpha=1.0, g=g)
    w = np.linalg.eigvalsh(B.dense())
    assert w.min() > 0
    assert np.linalg.norm(H.dense() @ B.dense() - np.eye(n)) < 1e-6
    with pytest.raises(RuntimeError):
        oracle.LBFGS(n).push(np.ones(n), np.ones(n), Bs=np.ones(n))   # test_lbfgs.jl:220-240


def test_hermitian_vs_dense():
    """test_linop.jl:360-380 (real symmetric)."""
    rng = np.random.default_rng(5)
    n = 17
    A = rng.standard_normal((n, n))
    d = rng.standard_normal(n)
    L = np.tril(A, -1)
    Cm = L + L.T + np.diag(d)
    v = SV(n)
    res = oracle.hermitian_mul(np.full(n, np.nan), d, A, v, 1.0, 0.0)
    assert np.linalg.norm(res - Cm @ v) <= 1e-12 * np.linalg.norm(v) * n
    r0 = rng.standard_normal(n)
    res = oracle.hermitian_mul(r0.copy(), d, A, v, 3.0, -4.0)
    assert np.allclose(res, 3.0 * (Cm @ v) - 4.0 * r0, rtol=1e-12)


def test_kron_vs_numpy():
    """test_kron.jl:2-39."""
    rng = np.random.default_rng(6)
    A = rng.standard_normal((3, 5))
    B = rng.standard_normal((4, 2))
    K = np.kron(A, B)
    x = rng.standard_normal(K.shape[1])
    res = oracle.kron_mul(np.full(K.shape[0], np.nan), A, B, x, 1.0, 0.0)
    assert np.linalg.norm(res - K @ x, 1) <= 1e-12 * np.linalg.norm(K, 1)
    xt = rng.standard_normal(K.shape[0])
    r0 = rng.standard_normal(K.shape[1])
    res = oracle.kron_mul(r0.copy(), A, B, xt, 2.0, 3.0, trans=True)
    assert np.allclose(res, 2.0 * (K.T @ xt) + 3.0 * r0, rtol=1e-12)


def test_mixed_precision_f32_scalars_f64():
    """mul!(res32, op32, v32, 2.0, 3.0) is evaluated in Float64 per element."""
    rng = np.random.default_rng(7)
    n = 1000
    d, v, r = (rng.standard_normal(n).astype(np.float32) for _ in range(3))
    a, b = 2.0 / 3.0, 1.0 / 7.0
    got = oracle.diag_mul(r.copy(), d, v, a, b, flags=oracle.SCALARS_F64)
    want = ((a * d.astype(np.float64)) * v.astype(np.float64) + b * r.astype(np.float64)).astype(np.float32)
    assert np.array_equal(got, want)
    got32 = oracle.diag_mul(r.copy(), d, v, a, b)
    want32 = (np.float32(a) * d) * v + np.float32(b) * r
    assert np.array_equal(got32, want32)
    assert not np.array_equal(got, got32)
    # one Float32 and one Float64 scalar: each term in its own type, the sum in Float64, one rounding on store
    d64, v64, r64 = d.astype(np.float64), v.astype(np.float64), r.astype(np.float64)
    got_ab = oracle.diag_mul(r.copy(), d, v, a, b, flags=oracle.BETA_F64)        # α::Float32, β::Float64
    want_ab = (((np.float32(a) * d) * v).astype(np.float64) + b * r64).astype(np.float32)
    assert np.array_equal(got_ab, want_ab)
    got_ba = oracle.diag_mul(r.copy(), d, v, a, b, flags=oracle.ALPHA_F64)       # α::Float64, β::Float32
    want_ba = ((a * d64) * v64 + (np.float32(b) * r).astype(np.float64)).astype(np.float32)
    assert np.array_equal(got_ba, want_ba)
    assert len({got.tobytes(), got32.tobytes(), got_ab.tobytes(), got_ba.tobytes()}) == 4
    assert oracle.scalar_flags(np.float32, np.float32(1), 2.0) == oracle.BETA_F64
    assert oracle.scalar_flags(np.float32, 1.5, 2) == oracle.ALPHA_F64
    assert oracle.scalar_flags(np.float64, 1.5, 2.0) == 0


def test_eye_zeros_ones_quirks():
    res = np.full(5, 2.0)
    oracle.eye_mul(res, np.ones(3), 1.0, 3.0, n_min=3)              # tail gets beta itself (special-operators.jl:42)
    assert np.array_equal(res, [7.0, 7.0, 7.0, 3.0, 3.0])
    res = np.full(4, np.nan)
    assert np.array_equal(oracle.zeros_mul(res, 0.0), np.zeros(4))
    assert np.array_equal(oracle.zeros_mul(np.full(4, 2.0), 1.5), np.full(4, 3.0))
    assert np.array_equal(oracle.ones_mul(np.full(3, 1.0), np.array([1.0, 2.0, 3.0]), 2.0, 1.0), np.full(3, 13.0))


def test_kat_diagqn_push(kat):
    """test_diag.jl:75-106: B.d after one push! from d = [1,-1,1] (DiagonalPSB, DiagonalAndrei) and
    SpectralGradient(1.0, 3).d[1]; plus the weak secant equation s'Bs = s'y (:52-72) and the allocation
    test's operators applied (:108-124 builds them on rand(5))."""
    cs = [c for c in kat if c["kind"] == "diagqn_push"]
    assert len(cs) == 3
    for c in cs:
        s, y = np.array(c["s"]), np.array(c["y"])
        for kind in ("psb", "andrei"):
            B = oracle.DiagonalQN(kind, np.array(c["d0"])).push(s, y)
            assert np.linalg.norm(B.d - np.array(c["expect_" + kind])) <= 1e-10
            Bs = B.mul(np.empty(3), s)
            assert abs(np.dot(s, Bs) - np.dot(s, y)) <= 1e-10             # weak secant (:59,70)
        S = oracle.DiagonalQN("spectral", np.array([1.0])).push(s, y)
        assert abs(S.d[0] - c["expect_spectral"]) <= 1e-10
        assert np.array_equal(S.mul(np.empty(3), s), S.d[0] * s)
    with np.testing.assert_raises(ZeroDivisionError):
        oracle.DiagonalQN("psb", np.ones(3)).push(np.zeros(3), np.ones(3))
    rng = np.random.default_rng(5)
    y = rng.standard_normal(50)
    s = rng.standard_normal(50)
    if np.dot(s, y) < 0:
        y = -y
    B = oracle.DiagonalQN("bfgs", np.ones(50)).push(s, y)                  # :245-248: d = |y| * sum|y| / (s'y/|s|^2)
    assert np.allclose(B.d, np.abs(y) * np.abs(y).sum() / (np.dot(s, y) / np.dot(s, s)), rtol=1e-13)
    assert np.array_equal(B.reset().d, np.ones(50))


# ------------------------------------------------------------------------------------------ round 2 KATs
def _cx(a):
    a = np.array(a, dtype=np.float64)
    return a[:, 0] + 1j * a[:, 1]


def test_kat_complex_diag(kat):
    """test/test_linop.jl:308-318 on ComplexF64 inputs: D*u, transpose(D)*u, D'*u (conj.(d)) and the 5-arg mul!
    with Float64 scalars — bit-exact (dyadic rationals), in both ComplexF64 and ComplexF32."""
    cs = _by_kind(kat, "cdiag")
    assert len(cs) == 2
    for c in cs:
        for dt in (np.complex128, np.complex64):
            d, u, r0 = (_cx(c[k]).astype(dt) for k in ("d", "u", "res0"))
            one, zero = (complex(1), complex(0)) if dt == np.complex128 else (np.complex64(1), np.complex64(0))
            fl = oracle.scalar_flags(dt, one, zero)
            res = np.full(d.size, np.nan + 1j * np.nan, dtype=dt)            # beta == 0 must not read res
            assert np.array_equal(oracle.diag_mul(res, d, u, one, zero, flags=fl), _cx(c["expect_apply"]).astype(dt)), c["name"]
            res = np.full(d.size, np.nan + 1j * np.nan, dtype=dt)
            got = oracle.diag_mul(res, d, u, one, zero, flags=fl | oracle.CONJ_D)
            assert np.array_equal(got, _cx(c["expect_ctapply"]).astype(dt)), c["name"]
            a, b = c["alpha"], c["beta"]                                      # Float64 (Real) scalars
            got = oracle.diag_mul(r0.copy(), d, u, a, b, flags=oracle.scalar_flags(dt, a, b))
            assert np.array_equal(got, _cx(c["expect_mul5"]).astype(dt)), c["name"]


def test_kat_complex_householder(kat):
    """test/test_linop.jl:511-517 on ComplexF64 inputs: H*u = u - 2*dot(v,u)*v with dot conjugating v;
    transpose(H)*u through the conj sandwich of src/adjtrans.jl:193-204: conj(H * conj(u))."""
    cs = _by_kind(kat, "chouseholder")
    assert len(cs) == 2
    for c in cs:
        h, u = _cx(c["h"]), _cx(c["u"])
        want, want_t = _cx(c["expect_apply"]), _cx(c["expect_tapply"])
        res = np.full(h.size, np.nan + 1j * np.nan)
        got = oracle.householder_mul(res, h, u, complex(1), complex(0), flags=0)
        assert np.linalg.norm(got - want) <= 1e-12 * np.linalg.norm(want), c["name"]
        got_t = np.conj(oracle.householder_mul(np.full(h.size, np.nan + 1j * np.nan), h, np.conj(u), complex(1), complex(0)))
        assert np.linalg.norm(got_t - want_t) <= 1e-12 * np.linalg.norm(want_t), c["name"]


def test_kat_hermitian(kat):
    (c,) = _by_kind(kat, "hermitian")
    A, d, v, x = np.array(c["A"]), np.array(c["d"]), np.array(c["v"]), np.array(c["x"])
    got = oracle.hermitian_mul(np.full(c["n"], np.nan), d, A, v, 1.0, 0.0)
    assert np.linalg.norm(got - np.array(c["expect_apply"])) <= 1e-13 * np.linalg.norm(c["expect_apply"])
    got = oracle.hermitian_mul(np.array(c["res0"]), d, A, x, c["alpha"], c["beta"])
    assert np.linalg.norm(got - np.array(c["expect_mul5"])) <= 1e-13 * np.linalg.norm(c["expect_mul5"])


def test_kat_dense_issue80_literal(kat):
    """test/test_linop.jl:587-595 — the reference holds the numbers: A = [1 1; 1 0], mul!(y, LinearOperator(A), ones(2))
    == [2, 1]; the oracle's dense restatement (src/constructors.jl:19-29) must reproduce them exactly, in both modes."""
    (c,) = _by_kind(kat, "dense")
    A, x = np.array(c["A"]), np.array(c["x"])
    for trans in (False, True):
        got = oracle.gemv(np.full(2, np.nan), A, x, 1.0, 0.0, trans=trans)
        assert np.array_equal(got, np.array(c["expect_apply"]))
        got = oracle.gemv(np.array(c["res0"]), A, x, c["alpha"], c["beta"], trans=trans)
        assert np.array_equal(got, np.array(c["expect_mul5"]))


def test_kat_complex_hermitian_and_dense(kat):
    """test/test_linop.jl:360-370 on ComplexF64: H = opHermitian(real.(diag), tril(A,-1)); H*v, transpose(H)*v (the conj
    sandwich of src/adjtrans.jl:193-204 around the hermitian prod!), H'*v and a 5-arg mul! with complex α, β; the same
    Hermitian matrix C as a dense LinearOperator(C) in the four op modes of the complex GEMV."""
    cs = _by_kind(kat, "chermitian")
    assert len(cs) == 2
    for c in cs:
        n = c["n"]
        A = np.array([[complex(*e) for e in row] for row in c["A"]])
        Cm = np.array([[complex(*e) for e in row] for row in c["C"]])
        d, v, r0 = np.array(c["d"]), _cx(c["v"]), _cx(c["res0"])
        nan = np.full(n, np.nan + 1j * np.nan)
        want = _cx(c["expect_apply"])
        got = oracle.hermitian_mul(nan.copy(), d, A, v, complex(1), complex(0))
        assert np.linalg.norm(got - want) <= 1e-13 * np.linalg.norm(want), c["name"]
        got_t = np.conj(oracle.hermitian_mul(nan.copy(), d, A, np.conj(v), complex(1), complex(0)))
        assert np.linalg.norm(got_t - _cx(c["expect_tapply"])) <= 1e-13 * np.linalg.norm(want), c["name"]
        al, be = complex(*c["alpha"]), complex(*c["beta"])
        got5 = oracle.hermitian_mul(r0.copy(), d, A, v, al, be)
        assert np.linalg.norm(got5 - _cx(c["expect_mul5"])) <= 1e-13 * np.linalg.norm(_cx(c["expect_mul5"])), c["name"]
        got32 = oracle.hermitian_mul(np.zeros(n, np.complex64), d.astype(np.float32), A.astype(np.complex64), v.astype(np.complex64),
                                     np.complex64(1), np.complex64(0))
        assert np.linalg.norm(got32 - want) <= 1e-5 * np.linalg.norm(want)
        for mode, Mx in (("N", Cm), ("T", Cm.T), ("C", Cm.conj().T), ("J", Cm.conj())):
            g = oracle.gemv(nan.copy(), Cm, v, complex(1), complex(0), trans=mode)
            assert np.linalg.norm(g - Mx @ v) <= 1e-13 * np.linalg.norm(want), (c["name"], mode)
        assert np.linalg.norm(oracle.gemv(nan.copy(), Cm, v, complex(1), complex(0), trans="N") - want) <= 1e-13 * np.linalg.norm(want)


def test_kat_kron(kat):
    (c,) = _by_kind(kat, "kron")
    A, B, K = np.array(c["A"]), np.array(c["B"]), np.array(c["K"])
    assert np.array_equal(np.kron(A, B), K)                              # the dense model IS Base.kron
    nK = np.linalg.norm(K, 1)
    got = oracle.kron_mul(np.full(K.shape[0], np.nan), A, B, np.array(c["x"]), 1.0, 0.0)
    assert np.linalg.norm(got - np.array(c["expect_apply"]), 1) <= 1e-12 * nK
    got = oracle.kron_mul(np.full(K.shape[1], np.nan), A, B, np.array(c["xt"]), 1.0, 0.0, trans=True)
    assert np.linalg.norm(got - np.array(c["expect_tapply"]), 1) <= 1e-12 * nK
    got = oracle.kron_mul(np.array(c["res0"]), A, B, np.array(c["x"]), c["alpha"], c["beta"])
    assert np.linalg.norm(got - np.array(c["expect_mul5"]), 1) <= 1e-12 * nK


def test_kat_complex_kron(kat):
    """test/test_kron.jl:3-36, the Float64 A x ComplexF64 B pairing: T*x, transpose(T)*x, T'*x, 5-arg mul!."""
    (c,) = _by_kind(kat, "ckron")
    A = np.array(c["A"])
    B = np.array([[complex(*e) for e in row] for row in c["B"]])
    K = np.array([[complex(*e) for e in row] for row in c["K"]])
    assert np.array_equal(K, np.kron(A, B))
    normK = np.abs(K).sum(axis=0).max()
    x, xt, r0 = _cx(c["x"]), _cx(c["xt"]), _cx(c["res0"])
    nan = lambda k: np.full(k, np.nan + 1j * np.nan)
    assert np.abs(oracle.kron_mul(nan(K.shape[0]), A, B, x, complex(1), complex(0)) - _cx(c["expect_apply"])).sum() <= 1e-12 * normK
    assert np.abs(oracle.kron_mul(nan(K.shape[1]), A, B, xt, complex(1), complex(0), trans="T") - _cx(c["expect_tapply"])).sum() <= 1e-12 * normK
    assert np.abs(oracle.kron_mul(nan(K.shape[1]), A, B, xt, complex(1), complex(0), trans="C") - _cx(c["expect_ctapply"])).sum() <= 1e-12 * normK
    got = oracle.kron_mul(r0.copy(), A, B, x, complex(*c["alpha"]), complex(*c["beta"]))
    assert np.abs(got - _cx(c["expect_mul5"])).sum() <= 1e-12 * normK


def test_complex_mixed_scalars_and_eye_zeros():
    """ComplexF32 data with Float64 / ComplexF64 / Float32 scalars: every term in its own type (component-wise
    model in NumPy, exact because every product is formed in float64 and rounded as the oracle does)."""
    rng = np.random.default_rng(3)
    n = 257
    mk = lambda: (rng.integers(-64, 64, n) / 8 + 1j * rng.integers(-64, 64, n) / 8).astype(np.complex64)
    d, v, r = mk(), mk(), mk()
    a, b = np.float32(1.5), 0.75 - 0.5j                                 # Float32 α, ComplexF64 β
    got = oracle.diag_mul(r.copy(), d, v, a, b, flags=oracle.scalar_flags(np.complex64, a, b))
    t = (np.float32(1.5) * d) * v                                       # complex64 arithmetic, exact on dyadics
    want = (t.astype(np.complex128) + b * r.astype(np.complex128)).astype(np.complex64)
    assert np.array_equal(got, want)
    e = oracle.eye_mul(r.copy(), v, 2.0, 0.5, flags=oracle.scalar_flags(np.complex64, 2.0, 0.5) | oracle.TAIL_BETA)
    assert np.array_equal(e, (2.0 * v.astype(np.complex128) + 0.5 * r.astype(np.complex128)).astype(np.complex64))
    z = oracle.zeros_mul(r.copy(), 0.5 + 0.25j, flags=oracle.scalar_flags(np.complex64, 0, 0.5 + 0.25j))
    assert np.array_equal(z, (r.astype(np.complex128) * (0.5 + 0.25j)).astype(np.complex64))
    assert np.array_equal(oracle.zeros_mul(np.full(4, np.nan + 0j), 0.0), np.zeros(4, dtype=np.complex128))


@pytest.mark.parametrize("npd", [np.float64, np.float32])
def test_sparse_csc_restatement_vs_an_independent_dense_product(npd):
    """oracle.csc_mul restates the SparseArrays loops behind `mul!(res, M::SparseMatrixCSC, v, α, β)`
    (src/constructors.jl:19-29). The reference holds no literal vectors for sparse operators — its tests compare them
    with dense matrices to sqrt(eps) (test/test_linop.jl:743-756, test/test_kron.jl:3-36) — so the restatement is pinned
    the same way, against a dense product formed independently (float64 / longdouble), for A*v and Aᵀ*v, β = 0 on NaN
    garbage, β = 1, general α, β, empty columns, unsorted rows and duplicate entries, Float32 data with Float64 scalars."""
    import scipy.sparse as sp
    rng = np.random.default_rng(99)
    eps = np.finfo(npd).eps
    for m, n, dens in ((1, 1, 1.0), (6, 9, 0.4), (40, 25, 0.15), (200, 300, 0.03), (30, 30, 0.0)):
        A = sp.random(m, n, dens, format="csc", random_state=int(rng.integers(1 << 30))).astype(npd)
        # shuffle the rows inside every column and duplicate one entry: legal input for the reference's loops
        cp, rv, nz = A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data.copy()
        for j in range(n):
            p = rng.permutation(cp[j + 1] - cp[j]) + cp[j]
            rv[cp[j]:cp[j + 1]], nz[cp[j]:cp[j + 1]] = rv[p], nz[p]
        if nz.size:
            j = int(np.searchsorted(cp, 0, side="right") - 1)
            rv = np.insert(rv, cp[j], rv[cp[j]]); nz = np.insert(nz, cp[j], npd(0.5)); cp[j + 1:] += 1
        D = np.zeros((m, n), np.longdouble)
        for j in range(n):
            for k in range(cp[j], cp[j + 1]):
                D[rv[k], j] += np.longdouble(nz[k])
        for trans in (False, True):
            Dm = D.T if trans else D
            v = rng.uniform(-1, 1, Dm.shape[1]).astype(npd)
            r0 = rng.uniform(-1, 1, Dm.shape[0]).astype(npd)
            for a, b, flags in ((1.0, 0.0, 0), (2.0, -3.0, 0), (-0.5, 1.0, 0), (1.25, 0.75, 0x1 | 0x8)):
                start = np.full(r0.size, np.nan, npd) if b == 0 else r0.copy()
                got = oracle.csc_mul(start, cp + 1, rv + 1, nz, m, n, v, a, b, trans=trans, flags=flags)
                want = a * (Dm @ v.astype(np.longdouble)) + (b * r0.astype(np.longdouble) if b != 0 else 0)
                scale = abs(a) * (np.abs(Dm) @ np.abs(v.astype(np.longdouble))).max(initial=0) + abs(b) + 1e-300
                assert np.isfinite(got).all()
                assert np.abs(got.astype(np.longdouble) - want).max() <= 16 * max(m, n) ** 0.5 * eps * scale, (m, n, trans, a, b)


@pytest.mark.parametrize("cdt", [np.complex128, np.complex64])
def test_complex_sparse_csc_restatement_vs_an_independent_dense_product(cdt):
    """The Complex{R} instantiation of oracle.csc_mul (SparseArrays `_spmatmul!` / `_At_or_Ac_mul_B!` with tfun =
    transpose or adjoint) against a dense product in extended precision: A*v, transpose(A)*v, A'*v, Real and Complex α, β,
    β = 0 on NaN, β = 1."""
    import scipy.sparse as sp
    rng = np.random.default_rng(5)
    eps = np.finfo(cdt).eps
    for m, n, dens in ((1, 1, 1.0), (9, 6, 0.4), (120, 200, 0.05)):
        A = sp.random(m, n, dens, format="csc", random_state=int(rng.integers(1 << 30))).astype(np.complex128)
        A.data = rng.standard_normal(A.nnz) + 1j * rng.standard_normal(A.nnz)
        A = A.astype(cdt)
        D = A.toarray().astype(np.clongdouble)
        for mode, Dm in ((False, D), ("T", D.T), ("C", D.conj().T)):
            v = (rng.standard_normal(Dm.shape[1]) + 1j * rng.standard_normal(Dm.shape[1])).astype(cdt)
            r0 = (rng.standard_normal(Dm.shape[0]) + 1j * rng.standard_normal(Dm.shape[0])).astype(cdt)
            for a, b in ((1.0, 0.0), (2.0, -3.0), (1.5 - 0.5j, 0.25 + 2j), (-0.5j, 1.0)):
                flags = (0x20 if not isinstance(a, complex) else 0) | (0x40 if not isinstance(b, complex) else 0) | 0x1 | 0x8
                start = np.full(r0.size, np.nan + 0j, cdt) if b == 0 else r0.copy()
                got = oracle.csc_mul(start, A.indptr + 1, A.indices + 1, A.data, m, n, v, a, b, trans=mode, flags=flags)
                want = a * (Dm @ v.astype(np.clongdouble)) + (b * r0.astype(np.clongdouble) if b != 0 else 0)
                scale = abs(a) * (np.abs(Dm) @ np.abs(v.astype(np.clongdouble))).max(initial=0) + abs(b) * np.abs(r0).max() + 1e-300
                assert np.isfinite(got).all()
                assert np.abs(got.astype(np.clongdouble) - want).max() <= 32 * max(m, n) ** 0.5 * eps * scale, (m, n, mode, a, b)
