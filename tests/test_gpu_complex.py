"""-m gpu: ComplexF64 / ComplexF32 instantiation of the elementwise leaves, opHouseholder, restriction/extension and
the Adjoint/Transpose/Conjugate wrapper routing (src/adjtrans.jl:90-261), through the C ABI via the host mirror.

The reference's own tests of opDiagonal and opHouseholder run on ComplexF64 (test/test_linop.jl:308-318, 511-517);
those two known-answer cases are restored here with their complex inputs. Tolerances: elementwise leaves BIT-EXACT
against the oracle's component-wise restatement of Julia's complex arithmetic; opHouseholder (one conjugated dot,
fixed-order tree) 1e-12 in ComplexF64 and 1e-5 in ComplexF32."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu

NPC = {torch.complex128: np.complex128, torch.complex64: np.complex64}
SIZES = [1, 2, 3, 7, 64, 255, 1000, 4097, 100_003, 1_048_577]


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def cx(a):
    a = np.array(a, dtype=np.float64)
    return a[:, 0] + 1j * a[:, 1]


def crand(rng, n, dt):
    return (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(dt)


def rel(a, b):
    nb = np.linalg.norm(b.astype(np.complex128))
    return np.linalg.norm(a.astype(np.complex128) - b.astype(np.complex128)) / (nb if nb else 1.0)


def same(a, b):
    """bit-identical, NaN payloads and signed zeros included"""
    return a.dtype == b.dtype and a.tobytes() == b.tobytes()


# ------------------------------------------------------------------------------------------ KATs
@pytest.mark.parametrize("dtype", [torch.complex128, torch.complex64])
def test_kat_complex_diag(lo, dev, kat, dtype):
    cs = [c for c in kat if c["kind"] == "cdiag"]
    assert len(cs) == 2
    npd = NPC[dtype]
    for c in cs:
        d, u, r0 = (cx(c[k]).astype(npd) for k in ("d", "u", "res0"))
        D = lo.opDiagonal(T(d, dev))
        assert lo.issymmetric(D) and not lo.ishermitian(D)            # hermitian = isreal(d): false for Complex{T}
        assert np.array_equal((D * T(u, dev)).cpu().numpy(), cx(c["expect_apply"]).astype(npd))
        assert np.array_equal((D.T * T(u, dev)).cpu().numpy(), cx(c["expect_tapply"]).astype(npd))
        assert np.array_equal((D.H * T(u, dev)).cpu().numpy(), cx(c["expect_ctapply"]).astype(npd))   # conj.(d)
        res = T(r0.copy(), dev)
        lo.mul(res, D, T(u, dev), c["alpha"], c["beta"])              # mul!(res, D, u, 2.0, 2.0)
        assert np.array_equal(res.cpu().numpy(), cx(c["expect_mul5"]).astype(npd))
        assert (lo.nprod(D), lo.ntprod(D), lo.nctprod(D)) == (3, 0, 1)   # transpose(D) is D itself (symmetric)


def test_kat_complex_householder(lo, dev, kat):
    cs = [c for c in kat if c["kind"] == "chouseholder"]
    assert len(cs) == 2
    for c in cs:
        h, u = cx(c["h"]), cx(c["u"])
        H = lo.opHouseholder(T(h, dev))
        assert lo.ishermitian(H) and not lo.issymmetric(H)            # symmetric = isreal(h)
        for op, key in ((H, "expect_apply"), (H.T, "expect_tapply"), (H.H, "expect_ctapply")):
            got = (op * T(u, dev)).cpu().numpy()
            want = cx(c[key])
            assert np.linalg.norm(got - want) <= 1e-12 * np.linalg.norm(want), (c["name"], key)


# ------------------------------------------------------------------------------------------ parity vs oracle
@pytest.mark.parametrize("dtype", [torch.complex128, torch.complex64])
@pytest.mark.parametrize("n", SIZES)
def test_complex_diag_bit_exact(lo, dev, dtype, n):
    rng = np.random.default_rng(4000 + n)
    npd = NPC[dtype]
    d, v, r0 = crand(rng, n, npd), crand(rng, n, npd), crand(rng, n, npd)
    D = lo.opDiagonal(T(d, dev))
    scal = [(complex(1), complex(0)), (0.7 - 0.3j, 1.25 + 0.5j), (2.0 / 3.0, 1.0 / 7.0), (1.5 + 0.25j, 0.0), (1, 0),
            (-1.25, 0.5 - 2j), (np.complex64(0.3 - 1j), np.float32(0.7)), (np.float32(1.1), 0.5 + 0.5j)]
    for alpha, beta in scal:
        for conj_d, op in ((False, D), (True, D.H)):
            res = T(r0.copy(), dev)
            if beta == 0:
                res.fill_(complex(float("nan"), float("nan")))       # beta == 0: res is never read
            lo.mul(res, op, T(v, dev), alpha, beta)
            fl = oracle.scalar_flags(npd, alpha, beta) | (oracle.CONJ_D if conj_d else 0)
            want = oracle.diag_mul(r0.copy(), d, v, alpha, beta, flags=fl)
            assert same(res.cpu().numpy(), want), (alpha, beta, conj_d)


@pytest.mark.parametrize("dtype", [torch.complex128, torch.complex64])
def test_complex_eye_zeros_scale_rect_diag(lo, dev, dtype):
    rng = np.random.default_rng(77)
    npd = NPC[dtype]
    S = lo.Storage(dtype, dev)
    for nrow, ncol in ((1000, 1000), (1031, 700), (700, 1031)):
        v, r0 = crand(rng, ncol, npd), crand(rng, nrow, npd)
        E = lo.opEye(dtype, nrow, ncol, S=S)
        Z = lo.opZeros(dtype, nrow, ncol, S=S)
        for alpha, beta in ((complex(1), complex(0)), (0.5 + 1j, 2.0 - 0.25j), (2.0, 3.0), (1.5 - 1j, 0.0)):
            fl = oracle.scalar_flags(npd, alpha, beta)
            res = T(r0.copy(), dev)
            lo.mul(res, E, T(v, dev), alpha, beta)
            assert same(res.cpu().numpy(), oracle.eye_mul(r0.copy(), v, alpha, beta, n_min=min(nrow, ncol), flags=fl | oracle.TAIL_BETA))
            res = T(r0.copy(), dev)
            lo.mul(res, Z, T(v, dev), alpha, beta)
            assert same(res.cpu().numpy(), oracle.zeros_mul(r0.copy(), beta, flags=fl))
        if nrow != ncol:                                              # rectangular opDiagonal: tail zeroed whatever β is
            d = crand(rng, min(nrow, ncol), npd)
            Dr = lo.opDiagonal(nrow, ncol, T(d, dev))
            for op, vin, nout, cj in ((Dr, v, nrow, False), (Dr.H, crand(rng, nrow, npd), ncol, True)):
                rr = crand(rng, nout, npd)
                res = T(rr.copy(), dev)
                lo.mul(res, op, T(vin, dev), 0.5 - 1j, 2.0)
                want = oracle.diag_mul(rr.copy(), d, vin[:d.size].copy(), 0.5 - 1j, 2.0, n_min=d.size,
                                       flags=oracle.scalar_flags(npd, 0.5 - 1j, 2.0) | (oracle.CONJ_D if cj else 0))
                assert same(res.cpu().numpy(), want)


@pytest.mark.parametrize("dtype", [torch.complex128, torch.complex64])
@pytest.mark.parametrize("n", SIZES)
def test_complex_householder_parity(lo, dev, dtype, n):
    rng = np.random.default_rng(5000 + n)
    npd = NPC[dtype]
    h = crand(rng, n, np.complex128)
    h = (h / np.linalg.norm(h)).astype(npd)
    v, r0 = crand(rng, n, npd), crand(rng, n, npd)
    H = lo.opHouseholder(T(h, dev))
    tol = 1e-12 if dtype == torch.complex128 else 1e-5
    for alpha, beta in ((complex(1), complex(0)), (2.0, -3.0), (0.5 - 1j, 0.25 + 2j)):
        res = T(r0.copy(), dev)
        if beta == 0:
            res.fill_(complex(float("nan"), float("nan")))
        lo.mul(res, H, T(v, dev), alpha, beta)
        want = oracle.householder_mul(r0.copy(), h, v, alpha, beta, flags=oracle.scalar_flags(npd, alpha, beta))
        assert rel(res.cpu().numpy(), want) <= tol, (n, alpha, beta)
    # involution (‖h‖ = 1): H(Hv) = v; and the conj sandwich: transpose(H) v = conj(H conj(v))
    w = H * (H * T(v, dev))
    assert rel(w.cpu().numpy(), v) <= 50 * tol
    t = (H.T * T(v, dev)).cpu().numpy()
    assert rel(t, np.conj((H * T(np.conj(v), dev)).cpu().numpy())) == 0.0
    a, b = H * T(v, dev), H * T(v, dev)
    assert torch.equal(a, b)                                          # fixed-order reduction: deterministic


@pytest.mark.parametrize("dtype", [torch.complex128, torch.complex64])
def test_complex_restriction_extension_bit_exact(lo, dev, dtype):
    """16-byte (ComplexF64) and 8-byte (ComplexF32) elements through the index kernels: pure data movement."""
    rng = np.random.default_rng(9)
    npd = NPC[dtype]
    n = 50_000
    v = crand(rng, n, npd)
    for idx in (rng.integers(1, n + 1, 7777), np.sort(rng.choice(n, 999, replace=False)) + 1):
        R = lo.opRestriction(idx.tolist(), n, device=dev)
        got = (R * T(v, dev)).cpu().numpy()
        assert same(got, v[idx - 1])
        u = crand(rng, idx.size, npd)
        want = np.zeros(n, dtype=npd)
        for k, i in enumerate(idx):
            want[i - 1] = u[k]
        assert same((R.H * T(u, dev)).cpu().numpy(), want)
    Rr = lo.opRestriction(lo.jrange(3, n - 5, 7), n, device=dev)
    assert same((Rr * T(v, dev)).cpu().numpy(), v[2:n - 5:7])


def test_conjugate_wrapper_reproduces_reference_quirk(lo, dev):
    """mul!(res, conj(A), v, α, β) = conj!(A*conj.(v)*α + β*res) — α, β and res are NOT conjugated first
    (src/adjtrans.jl:226-237): reproduced, not 'fixed'."""
    rng = np.random.default_rng(13)
    n = 513
    d, v, r0 = crand(rng, n, np.complex128), crand(rng, n, np.complex128), crand(rng, n, np.complex128)
    D = lo.opDiagonal(T(d, dev))
    a, b = 0.5 - 1.5j, 2.0 + 0.25j
    res = T(r0.copy(), dev)
    lo.mul(res, lo.conj(D), T(v, dev), a, b)
    want = np.conj(oracle.diag_mul(r0.copy(), d, np.conj(v), a, b, flags=oracle.scalar_flags(np.complex128, a, b)))
    assert same(res.cpu().numpy(), want)
    assert rel((lo.conj(D) * T(v, dev)).cpu().numpy(), np.conj(d) * v) <= 1e-15            # β = 0: the true conj(D) v


def test_complex_combinators_vs_dense(lo, dev):
    """compose / sum / scalar / cat over complex leaves against a dense NumPy model (test_linop.jl style)."""
    rng = np.random.default_rng(21)
    n = 300
    d1, d2 = crand(rng, n, np.complex128), crand(rng, n, np.complex128)
    h = crand(rng, n, np.complex128)
    h /= np.linalg.norm(h)
    D1, D2, H = lo.opDiagonal(T(d1, dev)), lo.opDiagonal(T(d2, dev)), lo.opHouseholder(T(h, dev))
    Hd = np.eye(n) - 2 * np.outer(h, np.conj(h))
    z = 0.5 - 2j
    op = (H * D1 + D2) * z - H
    M = (Hd @ np.diag(d1) + np.diag(d2)) * z - Hd
    v = crand(rng, n, np.complex128)
    for o, Md in ((op, M), (op.T, M.T), (op.H, M.conj().T)):
        assert rel((o * T(v, dev)).cpu().numpy(), Md @ v) <= 1e-12
    r0 = crand(rng, n, np.complex128)
    res = T(r0.copy(), dev)
    lo.mul(res, op, T(v, dev), 2.0 - 1j, 0.5 + 0.5j)
    assert rel(res.cpu().numpy(), (2.0 - 1j) * (M @ v) + (0.5 + 0.5j) * r0) <= 1e-12
    V = lo.vcat(D1, H)
    assert rel((V.H * T(np.concatenate([v, r0]), dev)).cpu().numpy(), np.conj(d1) * v + Hd.conj().T @ r0) <= 1e-12
    assert not lo.ishermitian(D1 * z) and lo.ishermitian(H * 2.0)
    assert lo.eltype(lo.opDiagonal(T(np.ones(4), dev)) * (1 + 1j)) == torch.complex128   # promote_type(Float64, ComplexF64)


def test_complex_diag_full_size_properties(lo, dev):
    """ComplexF64 opDiagonal at n = 5e7 (the same 2.4 GB the real headline moves at n = 1e8): linearity in v and
    bit-exact parity with the oracle on a strided sample."""
    n = 50_000_000
    g = torch.Generator(device=dev).manual_seed(5)
    def rnd():
        return torch.view_as_complex(torch.rand(n, 2, dtype=torch.float64, device=dev, generator=g) * 2 - 1)
    d, v1, v2 = rnd(), rnd(), rnd()
    D = lo.opDiagonal(d)
    r1, r2, r12 = D * v1, D * v2, D * (v1 + v2)
    assert (torch.linalg.vector_norm(r12 - (r1 + r2)) / torch.linalg.vector_norm(r12)).item() <= 1e-15
    idx = torch.arange(0, n, 99_991, device=dev)
    want = oracle.diag_mul(np.empty(idx.numel(), dtype=np.complex128), d[idx].cpu().numpy(), v1[idx].cpu().numpy(),
                           complex(1), complex(0))
    assert same(r1[idx].cpu().numpy(), want)


def test_real_only_leaves_reject_complex(lo, dev):
    """Not instantiated for complex element types: the quasi-Newton operators (real by construction in the reference) and
    block-diagonal fusion of complex dense blocks — loud TypeError, never a silent fallback."""
    with pytest.raises(TypeError):
        lo.LBFGSOperator(torch.complex128, 8, device=dev)


# ------------------------------------------------------------------------------------------ dense complex leaves
def cmat(rng, m, n, dt):
    return (rng.standard_normal((m, n)) + 1j * rng.standard_normal((m, n))).astype(dt)


def test_kat_complex_hermitian(lo, dev, kat):
    """test/test_linop.jl:360-370 with its ComplexF64 inputs: H = opHermitian(real.(diag(A)), tril(A,-1)); H*v == C*v,
    transpose(H)*v == transpose(C)*v, H'*v == C*v, 5-arg mul! with complex α, β; C itself as LinearOperator(C)."""
    cs = [c for c in kat if c["kind"] == "chermitian"]
    assert len(cs) == 2
    for c in cs:
        A = np.array([[complex(*e) for e in row] for row in c["A"]])
        Cm = np.array([[complex(*e) for e in row] for row in c["C"]])
        d, v, r0 = np.array(c["d"]), cx(c["v"]), cx(c["res0"])
        At = T(A.T.copy(), dev).t()                                   # column-major storage of A
        H = lo.opHermitian(T(d, dev), At)
        assert lo.ishermitian(H) and not lo.issymmetric(H)
        want, want_t = cx(c["expect_apply"]), cx(c["expect_tapply"])
        assert rel((H * T(v, dev)).cpu().numpy(), want) <= 1e-13
        assert rel((H.T * T(v, dev)).cpu().numpy(), want_t) <= 1e-13
        assert rel((H.H * T(v, dev)).cpu().numpy(), want) <= 1e-13
        res = T(r0.copy(), dev)
        lo.mul(res, H, T(v, dev), complex(*c["alpha"]), complex(*c["beta"]))
        assert rel(res.cpu().numpy(), cx(c["expect_mul5"])) <= 1e-13
        for M in (T(Cm, dev), T(Cm.T.copy(), dev).t()):               # row-major alias and column-major storage
            op = lo.LinearOperatorFromMatrix(M)
            assert rel((op * T(v, dev)).cpu().numpy(), want) <= 1e-13
            assert rel((op.T * T(v, dev)).cpu().numpy(), want_t) <= 1e-13
            assert rel((op.H * T(v, dev)).cpu().numpy(), want) <= 1e-13        # C is Hermitian


@pytest.mark.parametrize("dtype,tol", [(torch.complex128, 1e-12), (torch.complex64, 3e-5)])
@pytest.mark.parametrize("m,n", [(1, 1), (3, 2), (65, 7), (257, 300), (1000, 777), (31, 2049),
                                 (2048, 1030), (1030, 2048), (4096, 1024), (1024, 4096), (8192, 1100)])   # the last five: single-launch row bands (cgemv_rows_band_kernel), plain and conjugated
def test_complex_dense_gemv_all_modes_both_layouts(lo, dev, dtype, tol, m, n):
    """LinearOperator(M) on complex data: M*v, transpose(M)*u, M'*w with real / complex α, β (β = 0 must not read res),
    for column-major storage and for torch's row-major default (aliased: N/T swapped, M' through conj(M)*w)."""
    dt = NPC[dtype]
    rng = np.random.default_rng(m * 1000 + n)
    Mh = cmat(rng, m, n, dt)
    v, u = crand(rng, n, dt), crand(rng, m, dt)
    for layout in ("col", "row"):
        M = T(Mh, dev) if layout == "row" else T(Mh.T.copy(), dev).t()
        op = lo.LinearOperatorFromMatrix(M)
        for (a, b) in ((complex(1), complex(0)), (2.0, -3.0), (1.5 - 0.5j, 0.25 + 2j), (np.float32(0.5), 1j)):
            fl = oracle.scalar_flags(dt, a, b)
            for o, x, mode, nr in ((op, v, "N", m), (op.T, u, "T", n), (op.H, u, "C", n)):
                r0 = crand(rng, nr, dt)
                if b == 0:
                    r0[:] = np.nan + 1j * np.nan
                res = T(r0.copy(), dev)
                lo.mul(res, o, T(x, dev), a, b)
                want = oracle.gemv(r0.copy() if b != 0 else np.zeros(nr, dt), Mh, x, a, b, trans=mode, flags=fl)
                assert rel(res.cpu().numpy(), want) <= tol, (layout, mode, a, b)
    # the matrix is aliased: an in-place update is seen
    M = T(Mh, dev)
    op = lo.LinearOperatorFromMatrix(M)
    M.mul_(2.0)
    lo.touched(M)
    assert rel((op * T(v, dev)).cpu().numpy(), 2.0 * (Mh.astype(np.complex128) @ v)) <= tol


@pytest.mark.parametrize("dtype,tol", [(torch.complex128, 1e-12), (torch.complex64, 3e-5)])
@pytest.mark.parametrize("n", [1, 2, 5, 64, 257, 1000, 2051])
def test_complex_hermitian_parity(lo, dev, dtype, tol, n):
    """opHermitian(d, A) on complex A with a real and with a complex diagonal against the oracle's statement-order
    restatement and against the dense Hermitian matrix; transpose / adjoint through the wrapper routing; views with a
    leading dimension."""
    dt = NPC[dtype]
    rdt = np.float64 if dt == np.complex128 else np.float32
    rng = np.random.default_rng(n)
    big = cmat(rng, n + 3, n + 2, dt)                                   # A is a view into a larger column-major array
    bigd = T(big.T.copy(), dev).t()
    Ad = bigd[1:n + 1, 0:n]
    A = big[1:n + 1, 0:n]
    v = crand(rng, n, dt)
    L = np.tril(A.astype(np.complex128), -1)
    for d in (rng.standard_normal(n).astype(rdt), crand(rng, n, dt)):
        H = lo.opHermitian(T(d, dev), Ad)
        Cm = L + L.conj().T + np.diag(d.astype(np.complex128))
        for (a, b) in ((complex(1), complex(0)), (2.0, -3.0), (1.5 - 0.5j, 0.25 + 2j)):
            r0 = crand(rng, n, dt)
            res = T(r0.copy(), dev)
            lo.mul(res, H, T(v, dev), a, b)
            want = oracle.hermitian_mul(r0.copy(), d, A, v, a, b, flags=oracle.scalar_flags(dt, a, b))
            assert rel(res.cpu().numpy(), want) <= tol, (n, a, b)
            dense = a * (Cm @ v.astype(np.complex128)) + b * r0.astype(np.complex128)
            assert rel(res.cpu().numpy(), dense) <= 10 * tol
        if np.isrealobj(d):                                             # Hermitian matrix: H' == H, transpose(H) == conj(H)
            assert rel((H.H * T(v, dev)).cpu().numpy(), Cm @ v) <= 10 * tol
            assert rel((H.T * T(v, dev)).cpu().numpy(), Cm.T @ v) <= 10 * tol
    Hc = lo.opHermitian(Ad)                                             # opHermitian(A): d = diag(A) (complex)
    Cm = L + L.conj().T + np.diag(np.diag(A).astype(np.complex128))
    assert rel((Hc * T(v, dev)).cpu().numpy(), Cm @ v) <= 10 * tol


@pytest.mark.parametrize("dtype,tol,n,aligned", [
    (torch.complex128, 1e-12, 129, True), (torch.complex128, 1e-12, 640, True),
    (torch.complex128, 1e-12, 3001, True),       # thin strips of 2 tiles, ragged last row group
    (torch.complex128, 1e-12, 4224, True),       # full-width strips (separate launches), every row group full
    (torch.complex128, 1e-12, 4301, True),       # ... with a ragged last row group
    (torch.complex64, 3e-5, 515, True), (torch.complex64, 3e-5, 515, False),
    (torch.complex64, 3e-5, 4200, True),         # 2-tile strips
    (torch.complex64, 3e-5, 4200, False),        # the same, masked loads (A not 16-byte aligned)
    (torch.complex64, 3e-5, 8500, True),         # full-width strips
])
def test_complex_hermitian_strip_regimes(lo, dev, dtype, tol, n, aligned):
    """The single-pass strip kernel of the complex opHermitian in each of its launch regimes (one merged launch with
    1- and 2-tile strips; separate launches with full-width strips; aligned and masked loads; ragged last row group)
    against the oracle, and against the two-pass form (rows, then columns) as an independent device implementation."""
    dt = NPC[dtype]
    rdt = np.float64 if dt == np.complex128 else np.float32
    rng = np.random.default_rng(7 * n + aligned)
    off = 0 if aligned else 1
    big = cmat(rng, n + 2 + off, n, dt)
    bigd = T(big.T.copy(), dev).t()
    Ad, A = bigd[off:n + off, :], big[off:n + off, :]
    v = crand(rng, n, dt)
    ctx = lo.get_ctx(dev)
    for d in (rng.standard_normal(n).astype(rdt), crand(rng, n, dt)):
        H = lo.opHermitian(T(d, dev), Ad)
        for (a, b) in ((complex(1), complex(0)), (1.5 - 0.5j, 0.25 + 2j)):
            r0 = crand(rng, n, dt)
            res = T(r0.copy(), dev)
            lo.mul(res, H, T(v, dev), a, b)
            want = oracle.hermitian_mul(r0.copy(), d, A, v, a, b, flags=oracle.scalar_flags(dt, a, b))
            assert rel(res.cpu().numpy(), want) <= tol, (n, a, b)
            ctx.tune("cherm_two_pass", 1)
            try:
                res2 = T(r0.copy(), dev)
                lo.mul(res2, H, T(v, dev), a, b)
            finally:
                ctx.tune("cherm_two_pass", 0)
            assert rel(res.cpu().numpy(), res2.cpu().numpy()) <= tol
    # structure: e_j picks column j of the Hermitian matrix (exact products with 0 and 1)
    for j in (0, n // 2, n - 1):
        e = np.zeros(n, dt)
        e[j] = 1
        H = lo.opHermitian(T(np.zeros(n, rdt), dev), Ad)
        col = (H * T(e, dev)).cpu().numpy()
        Lc = np.tril(A, -1)
        assert np.array_equal(col, (Lc[:, j] + Lc[j, :].conj()).astype(dt))


@pytest.mark.parametrize("dtype,tol", [(torch.complex128, 1e-12), (torch.float64, 1e-12), (torch.complex64, 3e-5)])
def test_mul_on_matrices_dense_and_diagonal(lo, dev, dtype, tol):
    """test/test_linop.jl:64-76 ("LinearOperator(Matrix)"): mv = hcat(v, -2v), mu = hcat(u, -2u);
    mul!(res_mat, op, mv), mul!(res_trans, transpose(op), mu), mul!(res_adj, op', mu) against A*mv, transpose(A)*mu, A'*mu
    (src/operations.jl:34-36, src/adjtrans.jl:139-156, 207-224, 251-261) — plus the 5-arg form, column-major and row-major
    matrix operands, and the other closures that are columnwise on matrices in the reference (opDiagonal, opEye, opZeros)."""
    cplx = dtype.is_complex
    npd = NPC[dtype] if cplx else np.float64
    rng = np.random.default_rng(5)
    nrow, ncol = 10, 6
    rd = lambda *sh: (rng.uniform(-1, 1, sh) + (1j * rng.uniform(-1, 1, sh) if cplx else 0)).astype(npd)
    A = rd(nrow, ncol)
    v = np.array([-(-1.0) ** i for i in range(1, ncol + 1)], dtype=npd)
    u = np.array([-(-1.0) ** i for i in range(1, nrow + 1)], dtype=npd)
    if cplx:
        v, u = v + 0.5j * v[::-1], u - 0.25j * u
    mv, mu = np.stack([v, -2 * v], axis=1), np.stack([u, -2 * u], axis=1)
    colmajor = lambda X: T(X.T.copy(), dev).t()
    for Ad in (colmajor(A), T(A, dev)):
        op = lo.LinearOperatorFromMatrix(Ad)
        for mk in (colmajor, lambda X: T(X, dev)):
            res = torch.empty(nrow, 2, dtype=dtype, device=dev)
            lo.mul(res, op, mk(mv))
            assert rel(res.cpu().numpy(), A @ mv) <= tol
            rt = mk(np.zeros((ncol, 2), npd))
            lo.mul(rt, lo.transpose(op), mk(mu))
            assert rel(rt.cpu().numpy(), A.T @ mu) <= tol
            ra = mk(np.zeros((ncol, 2), npd))
            lo.mul(ra, lo.adjoint(op), mk(mu))
            assert rel(ra.cpu().numpy(), A.conj().T @ mu) <= tol
            rc = mk(np.zeros((nrow, 2), npd))
            lo.mul(rc, lo.conj(op), mk(mv))
            assert rel(rc.cpu().numpy(), A.conj() @ mv) <= tol
            r0 = rd(nrow, 2)
            r5 = mk(r0.copy())
            lo.mul(r5, op, mk(mv), 2.0, -0.5)
            assert rel(r5.cpu().numpy(), 2.0 * (A @ mv) - 0.5 * r0) <= tol
        with pytest.raises(lo.LinearOperatorException, match="shape mismatch"):
            lo.mul(torch.empty(ncol, 3, dtype=dtype, device=dev), lo.transpose(op), colmajor(mu))
    d = rd(ncol)
    D = lo.opDiagonal(T(d, dev))
    r0 = rd(ncol, 2)
    res = colmajor(r0.copy())
    lo.mul(res, D, colmajor(mv), 2.0, 3.0)
    assert rel(res.cpu().numpy(), 2.0 * d[:, None] * mv + 3.0 * r0) <= tol
    lo.mul(res, lo.adjoint(D), colmajor(mv), 1.0, 0.0)
    assert rel(res.cpu().numpy(), d.conj()[:, None] * mv) <= tol
    E = lo.opEye(dtype, ncol, S=lo.Storage(dtype, dev))
    lo.mul(res, E, colmajor(mv), -1.0, 0.0)
    assert np.array_equal(res.cpu().numpy(), -mv)
    H = lo.opHouseholder(T(rd(ncol), dev))                     # dot(h, m) of the reference closure is not defined on matrices
    with pytest.raises(lo.LinearOperatorException, match="vectors only"):
        lo.mul(res, H, colmajor(mv))


def test_reference_basic_operations_testset(lo, dev):
    """test/test_linop.jl:7-226 "Basic operations" on A1 = a ComplexF64 10 x 6 matrix, statement by statement: the four
    ways of writing the same operator (LinearOperator(A1), LinearOperator(A1')', transpose(LinearOperator(transpose(A1))),
    conj(LinearOperator(conj(A1)))) — data type, size, errors, Matrix(op), unary +; LinearOperator(Matrix) products;
    op ± op, op ± matrix, op ± scalar, op × op over {A, transpose(A), A', conj(A)}², matrix × op, op × matrix,
    scalar × op, op × scalar."""
    nrow, ncol = 10, 6
    rng = np.random.default_rng(2024)
    rtol = float(np.sqrt(np.finfo(np.float64).eps))
    cm = lambda m, n: (rng.uniform(-1, 1, (m, n)) + 1j * rng.uniform(-1, 1, (m, n)))
    dm = lambda X: T(np.ascontiguousarray(X.T), dev).t()                       # column-major device matrix
    M = lambda op: lo.Matrix(op).cpu().numpy()
    sv = lambda n: np.array([-(-1.0) ** i for i in range(1, n + 1)], dtype=np.complex128)
    A1 = cm(nrow, ncol)
    forms = (lo.LinearOperatorFromMatrix(dm(A1)),
             lo.adjoint(lo.LinearOperatorFromMatrix(dm(A1.conj().T))),
             lo.transpose(lo.LinearOperatorFromMatrix(dm(A1.T))),
             lo.conj(lo.LinearOperatorFromMatrix(dm(A1.conj()))))
    for op in forms:
        assert lo.eltype(op) == torch.complex128
        assert lo.size(op) == (nrow, ncol) and lo.size(op, 1) == nrow and lo.size(op, 2) == ncol
        with pytest.raises(lo.LinearOperatorException):
            lo.size(op, 3)
        with pytest.raises(lo.LinearOperatorException):
            op * torch.ones(ncol + 1, dtype=torch.complex128, device=dev)
        assert lo.issymmetric(op) is False and lo.ishermitian(op) is False
        assert np.linalg.norm(A1 - M(op)) <= 1e-15 * np.linalg.norm(A1) * 10
        assert np.linalg.norm(A1 - M(+op)) <= 1e-15 * np.linalg.norm(A1) * 10
    # "LinearOperator(Matrix)"
    op = forms[0]
    assert np.linalg.norm(A1.T - M(lo.transpose(op))) <= rtol * np.linalg.norm(A1)
    assert np.linalg.norm(A1.conj().T - M(lo.adjoint(op))) <= rtol * np.linalg.norm(A1)
    v, u = sv(ncol), sv(nrow)
    assert np.linalg.norm(A1 @ v - (op * T(v, dev)).cpu().numpy()) <= rtol * np.linalg.norm(v)
    assert np.linalg.norm(A1.T @ u - (lo.transpose(op) * T(u, dev)).cpu().numpy()) <= rtol * np.linalg.norm(u)
    assert np.linalg.norm(A1.conj().T @ u - (lo.adjoint(op) * T(u, dev)).cpu().numpy()) <= rtol * np.linalg.norm(u)
    # "Basic arithmetic operations": op ± op, matrix ± op, op ± matrix
    B1 = cm(nrow, ncol)
    for sgn, name in ((1, "+"), (-1, "-")):
        C = A1 + sgn * B1
        LA, LB = lo.LinearOperatorFromMatrix(dm(A1)), lo.LinearOperatorFromMatrix(dm(B1))
        variants = [LA + LB if sgn > 0 else LA - LB, (LA + dm(B1)) if sgn > 0 else (LA - dm(B1))]
        if sgn > 0:
            variants.append(lo.LinearOperatorFromMatrix(dm(A1)) + LB)
        for opC in variants:
            assert np.linalg.norm((opC * T(v, dev)).cpu().numpy() - C @ v) <= rtol * np.linalg.norm(v), name
            assert np.linalg.norm((lo.transpose(opC) * T(u, dev)).cpu().numpy() - C.T @ u) <= rtol * np.linalg.norm(u)
            assert np.linalg.norm((lo.adjoint(opC) * T(u, dev)).cpu().numpy() - C.conj().T @ u) <= rtol * np.linalg.norm(u)
    # "Operator ± scalar" (src/operations.jl:222-234: x * opOnes(nrow, ncol))
    x = 2.12345
    for opC, want in ((forms[0] + x, A1 + x), (x + forms[0], A1 + x), (forms[0] - x, A1 - x), (x - forms[0], x - A1)):
        assert np.linalg.norm(want - M(opC)) <= rtol * np.linalg.norm(want)
    # "Operator × Operator"
    A4, B4 = cm(ncol, ncol), cm(ncol, ncol)
    four = lambda X: (X, X.T, X.conj().T, X.conj())
    v6 = sv(ncol)
    for Ai in four(A4):
        for Bi in four(B4):
            C = Ai @ Bi
            opC = lo.LinearOperatorFromMatrix(dm(Ai)) * lo.LinearOperatorFromMatrix(dm(Bi))
            assert np.linalg.norm((opC * T(v6, dev)).cpu().numpy() - C @ v6) <= rtol * np.linalg.norm(v6)
            assert np.linalg.norm((lo.transpose(opC) * T(v6, dev)).cpu().numpy() - C.T @ v6) <= rtol * np.linalg.norm(v6)
            assert np.linalg.norm((lo.adjoint(opC) * T(v6, dev)).cpu().numpy() - C.conj().T @ v6) <= rtol * np.linalg.norm(v6)
    B2 = cm(ncol, ncol + 1)
    C = A1 @ B2
    opC = lo.LinearOperatorFromMatrix(dm(A1)) * lo.LinearOperatorFromMatrix(dm(B2))
    v7 = sv(ncol + 1)
    assert np.linalg.norm((opC * T(v7, dev)).cpu().numpy() - C @ v7) <= rtol * np.linalg.norm(v7)
    assert np.linalg.norm((lo.transpose(opC) * T(u, dev)).cpu().numpy() - C.T @ u) <= rtol * np.linalg.norm(u)
    assert np.linalg.norm((lo.adjoint(opC) * T(u, dev)).cpu().numpy() - C.conj().T @ u) <= rtol * np.linalg.norm(u)
    with pytest.raises(lo.LinearOperatorException):
        lo.LinearOperatorFromMatrix(dm(A1)) + lo.LinearOperatorFromMatrix(dm(B2))
    with pytest.raises(lo.LinearOperatorException):
        lo.LinearOperatorFromMatrix(dm(B2)) * lo.LinearOperatorFromMatrix(dm(A1))
    # "Matrix × operator", "Operator × matrix", "Scalar × operator", "Operator × scalar"
    assert np.linalg.norm(C - M(lo.LinearOperatorFromMatrix(dm(A1)) * dm(B2))) <= rtol * np.linalg.norm(C)
    assert np.linalg.norm(C - M(lo.compose(lo.LinearOperatorFromMatrix(dm(A1)), lo.LinearOperatorFromMatrix(dm(B2))))) <= rtol * np.linalg.norm(C)
    AA1 = x * A1
    assert np.linalg.norm(AA1 - M(x * lo.LinearOperatorFromMatrix(dm(A1)))) <= rtol * np.linalg.norm(AA1)
    assert np.linalg.norm(AA1 - M(lo.LinearOperatorFromMatrix(dm(A1)) * x)) <= rtol * np.linalg.norm(AA1)


def test_complex_dense_in_operator_trees_and_contract(lo, dev):
    """complex dense leaves compose with the complex elementwise leaves (sum, product, cat) against dense NumPy, and a
    warmed apply issues launches only."""
    import ctypes as C
    rng = np.random.default_rng(12)
    n = 300
    Mh, dh, hh = cmat(rng, n, n, np.complex128), crand(rng, n, np.complex128), crand(rng, n, np.complex128)
    hh /= np.linalg.norm(hh)
    M, D, Hh = lo.LinearOperatorFromMatrix(T(Mh, dev)), lo.opDiagonal(T(dh, dev)), lo.opHouseholder(T(hh, dev))
    A = cmat(rng, n, n, np.complex128)
    Hm = lo.opHermitian(T(rng.standard_normal(n), dev), T(A.T.copy(), dev).t())
    op = (M * D + Hh.H * M.H) * (2 - 1j) + Hm
    dense = (Mh @ np.diag(dh) + (np.eye(n) - 2 * np.outer(hh, hh.conj())).conj().T @ Mh.conj().T) * (2 - 1j) + lo.Matrix(Hm).cpu().numpy()
    x = crand(rng, n, np.complex128)
    assert rel((op * T(x, dev)).cpu().numpy(), dense @ x) <= 1e-11
    assert rel((op.H * T(x, dev)).cpu().numpy(), dense.conj().T @ x) <= 1e-11
    res = torch.empty(n, dtype=torch.complex128, device=dev)
    xt = T(x, dev)
    for _ in range(3):
        lo.mul(res, op, xt, 1.0, 0.0)

    def snap():
        a = (C.c_int64 * 12)()
        lo._lib.call("mxlo_debug_counters", a)
        return list(a)
    torch.cuda.synchronize()
    s0 = snap()
    lo.mul(res, M, xt, 2.0, -1.0)
    lo.mul(res, M.H, xt, 2.0, -1.0)
    lo.mul(res, Hm, xt, 2.0, -1.0)
    s1 = snap()
    dlt = [b - a for a, b in zip(s0, s1)]
    assert dlt[10] >= 6 and not any(dlt[k] for k in range(12) if k != 10), dlt      # launches only


def test_real_operator_applied_to_complex_vectors(lo, dev):
    """test/test_kron.jl "issue110": K = kron(A, LinearOperator(A)) with Float64 A applied to a ComplexF64 x gives a
    ComplexF64 y. Generalised: any real operator (leaves, combinators, wrappers, quasi-Newton) on complex vectors equals
    the dense real matrix times the complex vector, 3-arg and 5-arg with complex / real scalars."""
    rng = np.random.default_rng(110)
    A = rng.random((2, 2))
    At = T(A, dev)
    K = lo.kron(At, lo.LinearOperatorFromMatrix(At))
    x = crand(rng, 4, np.complex128)
    y = K * T(x, dev)
    assert y.dtype == torch.complex128                               # @test eltype(y) == Complex{Float64}
    assert rel(y.cpu().numpy(), np.kron(A, A) @ x) <= 1e-13
    n = 257
    d, h = rng.standard_normal(n), rng.standard_normal(n)
    h /= np.linalg.norm(h)
    M = rng.standard_normal((n, n))
    B = lo.LBFGSOperator(n, mem=3, device=dev)
    for _ in range(4):
        s = rng.uniform(-1, 1, n)
        lo.push(B, T(s, dev), T(s * rng.uniform(0.5, 2.0, n), dev))
    op = lo.opDiagonal(T(d, dev)) * lo.opHouseholder(T(h, dev)) + lo.LinearOperatorFromMatrix(T(M, dev)).T * 0.5 + B
    dense = lo.Matrix(op).cpu().numpy()
    v = crand(rng, n, np.complex128)
    for o, Dm in ((op, dense), (op.T, dense.T), (op.H, dense.T)):
        assert rel((o * T(v, dev)).cpu().numpy(), Dm @ v) <= 1e-11
        r0 = crand(rng, n, np.complex128)
        res = T(r0.copy(), dev)
        lo.mul(res, o, T(v, dev), 1.5 - 2j, 0.25 + 1j)
        assert rel(res.cpu().numpy(), (1.5 - 2j) * (Dm @ v) + (0.25 + 1j) * r0) <= 1e-11
        res = T(np.full(n, np.nan + 1j * np.nan), dev)
        lo.mul(res, o, T(v, dev), 2.0, 0.0)                          # beta == 0 does not read res
        assert rel(res.cpu().numpy(), 2.0 * (Dm @ v)) <= 1e-11
    v32 = crand(rng, n, np.complex64)
    D32 = lo.opDiagonal(T(d.astype(np.float32), dev))
    assert rel((D32 * T(v32, dev)).cpu().numpy(), d.astype(np.float32) * v32) <= 1e-6
    with pytest.raises(TypeError):
        lo.opDiagonal(T(d, dev)) * T(v32, dev)                       # Float64 operator next to ComplexF32 vectors


# ------------------------------------------------------------------------------------------ random complex operator trees
class _CGen:
    """Random trees over the complex leaves (diagonal, identity, zeros, Householder, dense in either layout, Hermitian
    with a real diagonal, kron with a real and a complex factor) and the reference's combinators (+, -, *, real / complex
    scalar, hcat, vcat and ONE kind of lazy wrapper per tree), evaluated side by side on dense NumPy matrices.

    One wrapper kind per tree (`wrap` = "T" or "H") because the reference's ConjugateLinearOperator — what
    transpose(adjoint(X)) and adjoint(transpose(X)) reduce to — conjugates the INCOMING res and does not conjugate α, β
    (src/adjtrans.jl:226-237): inside a sum or a cat, where it receives β = 1, the reference's result is not the
    mathematical one. That quirk is reproduced on purpose and pinned by
    test_conjugate_wrapper_reproduces_reference_quirk; here the trees stay where the reference is mathematically right."""

    def __init__(self, lo, dev, seed):
        self.lo, self.dev, self.rng = lo, dev, np.random.default_rng(50_000 + seed)
        self.wrap = "T" if seed % 2 else "H"

    def c(self, *shape):
        return self.rng.standard_normal(shape) + 1j * self.rng.standard_normal(shape)

    def leaf(self, m, n):
        lo, dev, rng = self.lo, self.dev, self.rng
        kinds = ["dense_col", "dense_row", "zeros", "sparse_csc", "sparse_csr"]
        if m == n:
            kinds += ["diag", "eye", "householder", "hermitian"]
            if m % 2 == 0 and m >= 4:
                kinds.append("kron")
        k = kinds[rng.integers(len(kinds))]
        S = lo.Storage(torch.complex128, dev)
        if k == "dense_col":
            A = self.c(m, n)
            return lo.LinearOperatorFromMatrix(T(A.T.copy(), dev).t()), A, k
        if k == "dense_row":
            A = self.c(m, n)
            return lo.LinearOperatorFromMatrix(T(A, dev)), A, k
        if k in ("sparse_csc", "sparse_csr"):                       # native complex sparse leaf (mxlo_csc_mul_c); CSR = transposed alias
            import scipy.sparse as sp
            A = self.c(m, n) * (rng.random((m, n)) < rng.uniform(0.1, 0.7))
            Sm = sp.csc_matrix(A)
            M = torch.sparse_csc_tensor(torch.from_numpy(Sm.indptr.astype(np.int64)), torch.from_numpy(Sm.indices.astype(np.int64)),
                                        torch.from_numpy(Sm.data.astype(np.complex128)), size=A.shape).to(dev)
            return lo.LinearOperatorFromMatrix(M if k == "sparse_csc" else M.to_sparse_csr()), A, k
        if k == "zeros":
            return lo.opZeros(torch.complex128, m, n, S=S), np.zeros((m, n), complex), k
        if k == "diag":
            d = self.c(n)
            return lo.opDiagonal(T(d, dev)), np.diag(d), k
        if k == "eye":
            return lo.opEye(torch.complex128, n, S=S), np.eye(n, dtype=complex), k
        if k == "householder":
            h = self.c(n)
            h /= np.linalg.norm(h)
            return lo.opHouseholder(T(h, dev)), np.eye(n) - 2 * np.outer(h, h.conj()), k
        if k == "hermitian":
            A, d = self.c(n, n), rng.standard_normal(n)
            L = np.tril(A, -1)
            return lo.opHermitian(T(d, dev), T(A.T.copy(), dev).t()), L + L.conj().T + np.diag(d), k
        A, B = rng.standard_normal((2, 2)), self.c(m // 2, n // 2)
        return lo.kron(T(A, dev), T(B.T.copy(), dev).t()), np.kron(A, B), k

    def tree(self, m, n, depth):
        lo, rng = self.lo, self.rng
        if depth == 0:
            return self.leaf(m, n)
        c = rng.integers(7)
        if c in (0, 1):
            a, A, da = self.tree(m, n, depth - 1)
            b, B, db = self.tree(m, n, depth - 1)
            return (a + b, A + B, f"({da} + {db})") if c == 0 else (a - b, A - B, f"({da} - {db})")
        if c == 2:
            k = int(rng.integers(1, 9))
            a, A, da = self.tree(m, k, depth - 1)
            b, B, db = self.tree(k, n, depth - 1)
            return a * b, A @ B, f"({da} * {db})"
        if c == 3:
            x = complex(rng.uniform(-2, 2), rng.uniform(-2, 2)) if rng.integers(2) else float(rng.uniform(-2, 2))
            a, A, d = self.tree(m, n, depth - 1)
            return x * a, x * A, f"({x:.2f} * {d})"
        if c == 4:
            a, A, d = self.tree(n, m, depth - 1)
            return (a.T, A.T, d + ".T") if self.wrap == "T" else (a.H, A.conj().T, d + ".H")
        if c == 5 and n >= 2:
            k = int(rng.integers(1, n))
            a, A, da = self.tree(m, k, depth - 1)
            b, B, db = self.tree(m, n - k, depth - 1)
            return lo.hcat(a, b), np.hstack([A, B]), f"hcat({da}, {db})"
        if c == 6 and m >= 2:
            k = int(rng.integers(1, m))
            a, A, da = self.tree(k, n, depth - 1)
            b, B, db = self.tree(m - k, n, depth - 1)
            return lo.vcat(a, b), np.vstack([A, B]), f"vcat({da}, {db})"
        return self.leaf(m, n)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("MXLO_CFUZZ_SEEDS", "160"))))
def test_random_complex_operator_tree_vs_dense(lo, dev, seed):
    """op*v and the tree's wrapper direction (transpose(op)*w for "T" trees, op'*w for "H" trees), both also in the 5-arg
    form with complex (α, β), and Matrix(op), for random complex operator trees: the leaves' prod!/tprod!/ctprod!, the
    inferred transposes of hermitian operators through the conj sandwiches of src/adjtrans.jl:113-136,193-204, and how
    every combinator forwards α and β."""
    g = _CGen(lo, dev, seed)
    rng = g.rng
    hi = 13 if seed % 4 else 33
    m, n = int(rng.integers(1, hi)), int(rng.integers(1, hi))
    op, M, desc = g.tree(m, n, depth=int(rng.integers(1, 4)))
    assert op.shape == M.shape == (m, n), desc
    v, w = g.c(n), g.c(m)
    scale = max(np.linalg.norm(M, 2), 1.0) * 40          # trees multiply up to 3 levels of O(1..10) factors
    close = lambda got, want, vec: np.linalg.norm(got - want) <= 1e-10 * scale * max(np.linalg.norm(vec), 1.0) + 1e-10 * np.linalg.norm(want)
    back, Mb = (op.T, M.T) if g.wrap == "T" else (op.H, M.conj().T)
    assert close((op * T(v, dev)).cpu().numpy(), M @ v, v), desc
    assert close((back * T(w, dev)).cpu().numpy(), Mb @ w, w), desc
    assert np.linalg.norm(lo.Matrix(op).cpu().numpy() - M) <= 1e-10 * scale * max(m, n), desc
    a, b = 1.5 - 0.5j, 0.25 + 2j
    for o, Mx, x, nr in ((op, M, v, m), (back, Mb, w, n)):
        r0 = g.c(nr)
        res = T(r0.copy(), dev)
        lo.mul(res, o, T(x, dev), a, b)
        assert close(res.cpu().numpy(), a * (Mx @ x) + b * r0, np.concatenate([x, r0])), desc


def test_complex_scalar_times_lazy_wrappers(lo, dev):
    """src/adjtrans.jl:266-272: x*A' = (conj(x)*A)', x*transpose(A) = transpose(x*A), x*conj(A) = conj(conj(x)*A) — a
    complex scalar is conjugated when it moves inside an adjoint / conj wrapper (found by the random-tree test above)."""
    rng = np.random.default_rng(24)
    A = cmat(rng, 5, 7, np.complex128)
    op = lo.LinearOperatorFromMatrix(T(A, dev))
    x = -1.87 - 0.4j
    for wrapped, M in ((op.H, A.conj().T), (op.T, A.T), (lo.conj(op), A.conj())):
        v = crand(rng, M.shape[1], np.complex128)
        for scaled in (x * wrapped, wrapped * x):
            assert rel((scaled * T(v, dev)).cpu().numpy(), x * (M @ v)) <= 1e-13
            w = crand(rng, M.shape[0], np.complex128)
            assert rel((scaled.H * T(w, dev)).cpu().numpy(), np.conj(x) * (M.conj().T @ w)) <= 1e-13
            assert rel((scaled.T * T(w, dev)).cpu().numpy(), x * (M.T @ w)) <= 1e-13


def test_shifted_operator_reference_testsets(lo, dev):
    """test/test_shifted_operator.jl: "Real Symmetric", "Complex Non-Hermitian" (σ = 1 + 2im, adjoint(op) = H' + conj(σ) I),
    "Mutation (Updating Sigma)", "Mutation (Dynamic Hermitian Check)", "Strict Type Constraint", "Coverage & Utilities"."""
    rng = np.random.default_rng(31)
    n = 5
    Hd = rng.random((n, n)); Hd = Hd + Hd.T
    H = lo.LinearOperatorFromMatrix(T(Hd, dev), symmetric=True, hermitian=True)
    op = lo.ShiftedOperator(H, 2.0)
    Aref = Hd + 2.0 * np.eye(n)
    x, y0 = rng.random(n), rng.random(n)
    assert op.shape == (n, n) and lo.issymmetric(op) and lo.ishermitian(op) and op.data.σ == 2.0
    y = torch.zeros(n, dtype=torch.float64, device=dev)
    lo.mul(y, op, T(x, dev))
    assert np.allclose(y.cpu().numpy(), Aref @ x)
    y = T(y0.copy(), dev)
    lo.mul(y, op, T(x, dev), 0.5, -1.0)
    assert np.allclose(y.cpu().numpy(), 0.5 * (Aref @ x) - y0)
    assert np.allclose((op.T * T(x, dev)).cpu().numpy(), Aref.T @ x)
    # Complex Non-Hermitian
    Hc = rng.random((n, n)) + 1j * rng.random((n, n))
    opc = lo.ShiftedOperator(lo.LinearOperatorFromMatrix(T(Hc, dev)), 1.0 + 2.0j)
    assert not lo.issymmetric(opc) and not lo.ishermitian(opc)
    Ac = Hc + (1.0 + 2.0j) * np.eye(n)
    xc = rng.random(n) + 1j * rng.random(n)
    assert np.allclose((opc.H * T(xc, dev)).cpu().numpy(), Ac.conj().T @ xc)
    assert np.allclose((opc * T(xc, dev)).cpu().numpy(), Ac @ xc)
    assert np.allclose((opc.T * T(xc, dev)).cpu().numpy(), Ac.T @ xc)
    r0 = rng.random(n) + 1j * rng.random(n)
    res = T(r0.copy(), dev)
    lo.mul(res, opc.H, T(xc, dev), 0.5 - 1j, 2j)
    assert np.allclose(res.cpu().numpy(), (0.5 - 1j) * (Ac.conj().T @ xc) + 2j * r0)
    # Mutation (Updating Sigma)
    H3 = rng.random((3, 3))
    op3 = lo.ShiftedOperator(lo.LinearOperatorFromMatrix(T(H3, dev)), 1.0)
    ones = torch.ones(3, dtype=torch.float64, device=dev)
    y1 = (op3 * ones).cpu().numpy()
    op3.data.σ = 10.0
    y2 = (op3 * ones).cpu().numpy()
    assert not np.allclose(y1, y2) and np.allclose(y2, (H3 + 10.0 * np.eye(3)) @ np.ones(3))
    # Mutation (Dynamic Hermitian Check)
    Hh = rng.random((3, 3)) + 1j * rng.random((3, 3)); Hh = Hh + Hh.conj().T
    oph = lo.ShiftedOperator(lo.LinearOperatorFromMatrix(T(Hh, dev), symmetric=False, hermitian=True), 2.0)
    assert lo.ishermitian(oph)
    oph.data.σ = 2.0 + 1.0j
    assert not lo.ishermitian(oph)
    oph.data.σ = 3.0
    assert lo.ishermitian(oph)
    # Strict Type Constraint
    H32 = lo.LinearOperatorFromMatrix(T(rng.random((5, 5)).astype(np.float32), dev))
    op32 = lo.ShiftedOperator(H32, 1.0)
    assert op32.eltype == torch.float32 and isinstance(op32.data.σ, np.float32)
    assert (op32 * T(rng.random(5).astype(np.float32), dev)).dtype == torch.float32
    with pytest.raises(TypeError):
        lo.ShiftedOperator(H32, 1.0 + 2.0j)              # convert(Float32, 1 + 2im): InexactError
    # Coverage & Utilities
    Hu = lo.LinearOperatorFromMatrix(T(rng.random((n, n)), dev))
    opu = lo.ShiftedOperator(Hu, 2.0)
    yy = torch.zeros(n, dtype=torch.float64, device=dev)
    lo.mul(yy, opu.T, T(x, dev), 0.5, 1.0)
    lo.mul(yy, opu.T, T(x, dev), 0.0, 1.0)
    lo.mul(yy, lo.ShiftedOperator(Hu, 0.0).T, T(x, dev))
    assert lo.isallocated5(opu) and lo.storage_type(opu) == lo.storage_type(Hu)
    opu.nprod = 10
    lo.reset(opu)
    assert opu.nprod == 0


def test_reference_transpose_adjoint_and_issue109_testsets(lo, dev):
    """test/test_linop.jl:382-427 "Transpose and adjoint" (ComplexF64 operator from user closures with ctprod! = nothing,
    then with tprod! = nothing: the missing one is inferred through conj sandwiches), :598-604 unary / scalar operations on
    wrappers, :607-631 "Sum / Cat with Adjoint and Transpose" (issue #109) — Matrix(...) compared EXACTLY like the reference."""
    rng = np.random.default_rng(109)
    n = 10
    A = cmat(rng, n, n, np.complex128)
    v = np.array([-(-1.0) ** i for i in range(1, n + 1)], dtype=np.complex128)     # simple_vector(ComplexF64, n)
    res_init = v.copy()
    Ad = lo.LinearOperatorFromMatrix(T(A.T.copy(), dev).t())
    S = lo.Storage(torch.complex128, dev)
    prod = lambda res, x, a, b: lo.mul(res, Ad, x, a, b)
    tprod = lambda res, x, a, b: lo.mul(res, Ad.T, x, a, b)
    ctprod = lambda res, x, a, b: lo.mul(res, Ad.H, x, a, b)
    rtol = np.sqrt(np.finfo(float).eps)
    nv = np.linalg.norm(v)
    alpha, beta = 2.0, -3.0
    for t_, c_, check in ((tprod, None, "adjoint"), (None, ctprod, "transpose")):
        op = lo.LinearOperator(torch.complex128, n, n, False, False, prod, t_, c_, S=S)
        ap = lambda o: (o * T(v, dev)).cpu().numpy()
        assert np.linalg.norm(A.T @ v - ap(op.T)) <= rtol * nv
        assert np.linalg.norm(A.conj().T @ v - ap(op.H)) <= rtol * nv
        assert np.linalg.norm(A @ v - ap(op.T.T)) <= rtol * nv
        assert np.linalg.norm(A @ v - ap(op.H.H)) <= rtol * nv
        assert np.linalg.norm(A.conj() @ v - ap(op.H.T)) <= rtol * nv
        assert np.linalg.norm(A.conj() @ v - ap(op.T.H)) <= rtol * nv
        res = T(res_init.copy(), dev)
        if check == "adjoint":
            lo.mul(res, op.H, T(v, dev), alpha, beta)
            assert np.linalg.norm(alpha * (A.conj().T @ v) + beta * res_init - res.cpu().numpy()) <= rtol * nv
        else:
            lo.mul(res, op.T, T(v, dev), alpha, beta)
            assert np.linalg.norm(alpha * (A.T @ v) + beta * res_init - res.cpu().numpy()) <= rtol * nv
    # unary and scalar operations on Adjoint and Transpose operators (real 5 x 3)
    R = rng.random((5, 3))
    opR = lo.LinearOperatorFromMatrix(T(R, dev))
    Mx = lambda o: lo.Matrix(o).cpu().numpy()
    for adjtrans in (lambda o: o.H, lambda o: o.T):
        assert np.array_equal(Mx(adjtrans(-opR)), Mx(-adjtrans(opR)))
        assert np.array_equal(Mx(adjtrans(2 * opR)), Mx(2 * adjtrans(opR)))
    # issue #109: sums and cats with adjoint / transpose wrappers of a complex operator, and with raw matrices
    A3 = rng.random((3, 3)) + 1j * rng.random((3, 3))
    A3d = T(A3, dev)
    opA = lo.LinearOperatorFromMatrix(A3d)
    for wrap, dense in ((lambda o: o.H, A3.conj().T), (lambda o: o.T, A3.T)):
        w = wrap(opA)
        wd = T(np.ascontiguousarray(dense), dev)
        assert np.array_equal(Mx(w + opA), A3 + dense) and np.array_equal(Mx(opA + w), A3 + dense)
        assert np.array_equal(Mx(w + A3d), A3 + dense) and np.array_equal(Mx(A3d + w), A3 + dense)
        assert np.array_equal(Mx(lo.hcat(w, opA)), np.hstack([dense, A3]))
        assert np.array_equal(Mx(lo.vcat(w, opA)), np.vstack([dense, A3]))
        assert np.array_equal(Mx(lo.vcat(w, w)), np.vstack([dense, dense]))
        assert np.array_equal(Mx(lo.hcat(opA, w)), np.hstack([A3, dense]))
        assert np.array_equal(Mx(lo.vcat(opA, w)), np.vstack([A3, dense]))
        assert np.array_equal(Mx(lo.hcat(w, A3d)), np.hstack([dense, A3]))
        assert np.array_equal(Mx(lo.vcat(w, A3d)), np.vstack([dense, A3]))
        assert np.array_equal(Mx(lo.hcat(A3d, w)), np.hstack([A3, dense]))
        assert np.array_equal(Mx(lo.vcat(A3d, w)), np.vstack([A3, dense]))
        blk = lo.hvcat((2, 2), w, opA, opA, w) if hasattr(lo, "hvcat") else lo.vcat(lo.hcat(w, opA), lo.hcat(opA, w))
        assert np.array_equal(Mx(blk), np.block([[dense, A3], [A3, dense]]))
        del wd


def _closure_op(lo, dev, A, which):
    """LinearOperator{ComplexF64}(m, n, false, false, prod!, tprod!, ctprod!) built from 2-ARGUMENT closures over a dense
    matrix, as test/test_adjtrans.jl:43-51,98-106 and test/test_cat.jl:55-66 do; `which` says which of tprod!/ctprod!
    is `nothing`."""
    Ad = lo.LinearOperatorFromMatrix(T(A, dev))
    m, n = A.shape
    p2 = lambda y, x: lo.mul(y, Ad, x)
    t2 = None if which == "no_t" else (lambda y, x: lo.mul(y, Ad.T, x))
    c2 = None if which == "no_ct" else (lambda y, x: lo.mul(y, Ad.H, x))
    return lo.LinearOperator(torch.complex128, m, n, False, False, p2, t2, c2, S=lo.Storage(torch.complex128, dev))


@pytest.mark.parametrize("which", ["dense", "no_ct", "no_t"])
def test_reference_adjtrans_testsets(lo, dev, which):
    """test/test_adjtrans.jl: "Adjoint/Transpose/Conjugate", "Derived Adjoint" (ctprod! = nothing), "Derived Transpose"
    (tprod! = nothing) on a 5 x 3 ComplexF64 operator: wrapper algebra identities, Matrix(fop) == foo(A) EXACTLY,
    -fop, (2+3im)*fop, fop*(2+3im), products with complex AND real vectors."""
    rng = np.random.default_rng(5)
    A = rng.random((5, 3)) + 1j * rng.random((5, 3))
    opA = lo.LinearOperatorFromMatrix(T(A, dev)) if which == "dense" else _closure_op(lo, dev, A, which)
    aop, cop, top = lo.adjoint(opA), lo.conj(opA), lo.transpose(opA)
    Mx = lambda o: lo.Matrix(o).cpu().numpy()
    for foo, fop, fA in ((lo.adjoint, aop, A.conj().T), (lo.conj, cop, A.conj()), (lo.transpose, top, A.T)):
        assert type(foo(opA)) is type(fop) and foo(opA).parent is opA          # foo(opA) === fop
        assert np.array_equal(Mx(fop), fA)                                      # Matrix(fop) == foo(A)
        assert foo(fop) is opA
        assert np.array_equal(Mx(-fop), -fA)
        assert np.allclose(Mx((2 + 3j) * fop), (2 + 3j) * fA) and np.allclose(Mx(fop * (2 + 3j)), fA * (2 + 3j))
    assert isinstance(lo.adjoint(top), type(cop)) and lo.adjoint(top).parent is opA      # adjoint(topA) === copA
    assert isinstance(lo.adjoint(cop), type(top)) and isinstance(lo.conj(aop), type(top))
    assert isinstance(lo.conj(top), type(aop)) and isinstance(lo.transpose(cop), type(aop))
    assert isinstance(lo.transpose(aop), type(cop))
    v = rng.random(5) + 1j * rng.random(5)
    assert np.allclose((aop * T(v, dev)).cpu().numpy(), A.conj().T @ v, rtol=1e-14, atol=0)
    assert np.allclose((top * T(v, dev)).cpu().numpy(), A.T @ v, rtol=1e-14, atol=0)
    vr = rng.random(5)                                                          # REAL vectors through a complex operator
    assert np.allclose((aop * T(vr, dev)).cpu().numpy(), A.conj().T @ vr)
    assert np.allclose((top * T(vr, dev)).cpu().numpy(), A.T @ vr)
    v3 = rng.random(3) + 1j * rng.random(3)
    assert np.allclose((cop * T(v3, dev)).cpu().numpy(), A.conj() @ v3)
    v3r = rng.random(3)
    assert np.allclose((cop * T(v3r, dev)).cpu().numpy(), A.conj() @ v3r)
    assert (opA * T(v3r, dev)).dtype == torch.complex128


@pytest.mark.parametrize("three_args", [False, True])
def test_reference_cat_testsets(lo, dev, three_args):
    """test/test_cat.jl "Concatenation" / "Concatenation 3-args" on ComplexF64: [A B C] and [A; B; C] of operators (dense,
    or built from 2-argument closures) against the dense concatenation and LinearOperator(D): products, transpose, adjoint,
    the 5-arg forms with (α, β) = (3, -4), the shape-mismatch exception, and the two small identity / zero block cases."""
    rng = np.random.default_rng(6)
    rtol = np.sqrt(np.finfo(float).eps)
    sv = lambda k: np.array([-(-1.0) ** i for i in range(1, k + 1)], dtype=np.complex128)
    mk = (lambda M: _closure_op(lo, dev, M, "full")) if three_args else (lambda M: lo.LinearOperatorFromMatrix(T(M, dev)))
    for axis in (1, 0):
        shapes = ((100, 100), (100, 10), (100, 90)) if axis == 1 else ((100, 100), (10, 100), (90, 100))
        blocks = [cmat(rng, m, n, np.complex128) for m, n in shapes]
        D = np.concatenate(blocks, axis=axis)
        Do = (lo.hcat if axis == 1 else lo.vcat)(*[mk(B) for B in blocks])
        Do2 = lo.LinearOperatorFromMatrix(T(D, dev))
        rhs, rhs2 = sv(D.shape[1]), sv(D.shape[0])
        for o in (Do, Do2):
            assert np.linalg.norm((o * T(rhs, dev)).cpu().numpy() - D @ rhs) <= rtol * np.linalg.norm(D @ rhs)
            assert np.linalg.norm((o.T * T(rhs2, dev)).cpu().numpy() - D.T @ rhs2) <= rtol * np.linalg.norm(D.T @ rhs2)
            assert np.linalg.norm((o.H * T(rhs2, dev)).cpu().numpy() - D.conj().T @ rhs2) <= rtol * np.linalg.norm(D.conj().T @ rhs2)
            a, b = 3.0, -4.0
            for oo, Dm, x in ((o, D, rhs), (o.T, D.T, rhs2), (o.H, D.conj().T, rhs2)):
                r0 = rng.random(Dm.shape[0]) + 1j * rng.random(Dm.shape[0])
                res = T(r0.copy(), dev)
                lo.mul(res, oo, T(x, dev), a, b)
                want = a * (Dm @ x) + b * r0
                assert np.linalg.norm(res.cpu().numpy() - want) <= rtol * np.linalg.norm(want)
    ones55 = lo.LinearOperatorFromMatrix(torch.ones(5, 5, dtype=torch.float64, device=dev))
    eye3 = lo.opEye(torch.float64, 3, S=lo.Storage(torch.float64, dev))
    with pytest.raises(lo.LinearOperatorException):
        lo.hcat(ones55, eye3)
    with pytest.raises(lo.LinearOperatorException):
        lo.vcat(ones55, eye3)
    S64 = lo.Storage(torch.float64, dev)
    K = lo.hvcat((2, 2), lo.opEye(torch.float64, 2, S=S64), lo.opZeros(torch.float64, 2, 3, S=S64),
                 lo.opZeros(torch.float64, 3, 2, S=S64), lo.opEye(torch.float64, 3, S=S64))
    v5 = torch.tensor([1.0, -1.0, 1.0, -1.0, 1.0], dtype=torch.float64, device=dev)
    assert torch.equal(K * v5, v5)                                              # all(v .== K * v)
    K2 = lo.vcat(lo.opEye(torch.float64, 2, S=S64), torch.eye(2, dtype=torch.float64, device=dev))
    v2 = torch.tensor([1.0, -1.0], dtype=torch.float64, device=dev)
    assert torch.equal(K2 * v2, torch.cat([v2, v2]))


def test_reference_blockdiagonal_issue107_and_three_args_complex(lo, dev):
    """test/test_linop.jl:718-757 "BlockDiagonal" (issue #107) with ComplexF64 blocks given as operators and as matrices,
    and :768-797 "3-args" for Complex{Float64}: a 2-argument prod! with tprod! = nothing and a 2-argument ctprod!."""
    rng = np.random.default_rng(107)
    rtol = np.sqrt(np.finfo(float).eps)
    A, B, C = cmat(rng, 3, 4, np.complex128), cmat(rng, 3, 3, np.complex128), cmat(rng, 4, 2, np.complex128)
    D = np.zeros((10, 9), dtype=np.complex128)
    D[:3, :4], D[3:6, 4:7], D[6:, 7:] = A, B, C
    Mx = lambda o: lo.Matrix(o).cpu().numpy()
    for M in (lo.BlockDiagonalOperator(*[lo.LinearOperatorFromMatrix(T(X, dev)) for X in (A, B, C)]),
              lo.BlockDiagonalOperator(T(A, dev), T(B, dev), T(C, dev))):
        assert M.size(1) == 10 and M.size(2) == 9
        assert np.linalg.norm(Mx(M) - D) <= rtol * np.linalg.norm(D)
        assert np.linalg.norm(Mx(M.T) - D.T) <= rtol * np.linalg.norm(D)
        assert np.linalg.norm(Mx(M.H) - D.conj().T) <= rtol * np.linalg.norm(D)
    # 3-args, Complex{Float64}
    A12 = cmat(rng, 12, 10, np.complex128)
    b = crand(rng, 10, np.complex128)
    Ad = lo.LinearOperatorFromMatrix(T(A12, dev))
    opA = lo.LinearOperator(torch.complex128, 12, 10, False, False, lambda res, v: lo.mul(res, Ad, v), None,
                            lambda res, w: lo.mul(res, Ad.H, w), S=lo.Storage(torch.complex128, dev))
    assert not lo.has_args5(opA)
    assert rel((opA * T(b, dev)).cpu().numpy(), A12 @ b) <= 1e-14
    res = T(crand(rng, 12, np.complex128), dev)
    lo.mul(res, opA, T(b, dev))
    assert rel(res.cpu().numpy(), A12 @ b) <= 1e-14
    for alpha, beta in ((2.0, 3.0), (1.0, 3.0), (2.0, 0.0)):
        res2 = res.cpu().numpy().copy()
        lo.mul(res, opA, T(b, dev), alpha, beta)
        assert rel(res.cpu().numpy(), alpha * (A12 @ b) + beta * res2) <= 1e-13
        c, r3 = crand(rng, 12, np.complex128), crand(rng, 10, np.complex128)
        res3 = T(r3.copy(), dev)
        lo.mul(res3, opA.T, T(c, dev), alpha, beta)
        assert np.linalg.norm(alpha * (A12.T @ c) + beta * r3 - res3.cpu().numpy()) <= rtol
        assert rel((opA.T * T(c, dev)).cpu().numpy(), A12.T @ c) <= 1e-13
        r4 = res3.cpu().numpy().copy()
        lo.mul(res3, opA.H, T(c, dev), alpha, beta)
        assert np.linalg.norm(alpha * (A12.conj().T @ c) + beta * r4 - res3.cpu().numpy()) <= rtol
        assert rel((opA.H * T(c, dev)).cpu().numpy(), A12.conj().T @ c) <= 1e-13


def test_reference_identity_ones_zeros_on_complex_vectors(lo, dev):
    """test/test_linop.jl:229-306 "Identity", "Ones", "Zeros": the reference builds these with their DEFAULT element type
    (Float64) and applies them to simple_vector(ComplexF64, n) — a real operator on complex vectors."""
    nrow, ncol = 10, 6
    eps_ = np.finfo(float).eps
    sv = lambda k: np.array([-(-1.0) ** i for i in range(1, k + 1)], dtype=np.complex128)
    S = lo.Storage(torch.float64, dev)
    v = sv(nrow)
    for opI in (lo.opEye(nrow, S=S), lo.opEye(nrow, nrow, S=S)):
        for o in (opI, opI.T, opI.H):
            assert np.linalg.norm((o * T(v, dev)).cpu().numpy() - v) <= eps_ * np.linalg.norm(v)
        assert np.array_equal(lo.Matrix(opI).cpu().numpy(), np.eye(nrow))
    w = lo.opEye(nrow, S=S) * T(v, dev)
    w[0] = -1.0
    assert v[0] != w[0].item()                                           # the product is a fresh vector
    opI = lo.opEye(nrow, ncol, S=S)
    vc = sv(ncol)
    v0, vu = np.concatenate([vc, np.zeros(nrow - ncol)]), np.concatenate([vc, np.ones(nrow - ncol)])
    assert np.linalg.norm((opI * T(vc, dev)).cpu().numpy() - v0) <= eps_ * np.linalg.norm(vc)
    assert np.linalg.norm((opI.T * T(vu, dev)).cpu().numpy() - vc) <= eps_ * np.linalg.norm(vc)
    assert np.linalg.norm((opI.H * T(vu, dev)).cpu().numpy() - vc) <= eps_ * np.linalg.norm(vc)
    assert np.array_equal(lo.Matrix(opI).cpu().numpy(), np.eye(nrow, ncol))
    opI = lo.opEye(ncol, nrow, S=S)
    assert np.linalg.norm((opI * T(vu, dev)).cpu().numpy() - vc) <= eps_ * np.linalg.norm(vc)
    assert np.linalg.norm((opI.T * T(vc, dev)).cpu().numpy() - v0) <= eps_ * np.linalg.norm(vc)
    assert np.linalg.norm((opI.H * T(vc, dev)).cpu().numpy() - v0) <= eps_ * np.linalg.norm(vc)
    rtol = np.sqrt(eps_)
    E = lo.opOnes(nrow, ncol, S=S)
    u = sv(ncol)
    assert np.linalg.norm((E * T(u, dev)).cpu().numpy() - u.sum() * np.ones(nrow)) <= rtol * np.linalg.norm(u)
    assert np.linalg.norm((E.T * T(v, dev)).cpu().numpy() - v.sum() * np.ones(ncol)) <= rtol * np.linalg.norm(v)
    assert np.linalg.norm((E.H * T(v, dev)).cpu().numpy() - v.sum() * np.ones(ncol)) <= rtol * np.linalg.norm(v)
    O = lo.opZeros(nrow, ncol, S=S)
    assert np.linalg.norm((O * T(u, dev)).cpu().numpy()) <= eps_
    assert np.linalg.norm((O.T * T(v, dev)).cpu().numpy()) <= eps_ and np.linalg.norm((O.H * T(v, dev)).cpu().numpy()) <= eps_
