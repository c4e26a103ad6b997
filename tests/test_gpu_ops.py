"""-m gpu: matrix-carrying leaves (opHermitian, kron, dense LinearOperator, BlockDiagonalOperator) and
the combinators / cat, compared with the oracle and with independent dense NumPy models, the way the
reference's tests do (test_linop.jl, test_cat.jl, test_kron.jl, test_adjtrans.jl)."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
NP = {torch.float64: np.float64, torch.float32: np.float32}
SV = lambda n: np.array([-(-1.0) ** i for i in range(1, n + 1)])


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def TM(a, dev):
    """column-major device matrix (what a Julia Matrix is)."""
    return torch.from_numpy(np.asfortranarray(a).T.copy()).to(dev).t()


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    nb = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (nb if nb else 1.0)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("m,n", [(10, 6), (6, 10), (1, 1), (300, 257), (1025, 33), (64, 2000)])
def test_dense_operator(lo, dev, dtype, m, n):
    """test_linop.jl:7-226 style: op*v, transpose(op)*u, op'*u, 5-arg forms vs the dense matrix."""
    rng = np.random.default_rng(m * 1000 + n)
    npd = NP[dtype]
    A = rng.standard_normal((m, n)).astype(npd)
    op = lo.LinearOperatorFromMatrix(TM(A, dev))
    v, u = rng.standard_normal(n).astype(npd), rng.standard_normal(m).astype(npd)
    tol = 1e-12 if dtype == torch.float64 else 2e-5
    assert op.shape == (m, n) and lo.size(op, 1) == m and lo.size(op.T, 1) == n
    assert rel((op * T(v, dev)).cpu().numpy(), A @ v) <= tol
    assert rel((op.T * T(u, dev)).cpu().numpy(), A.T @ u) <= tol
    assert rel((op.H * T(u, dev)).cpu().numpy(), A.T @ u) <= tol
    r0 = rng.standard_normal(m).astype(npd)
    res = T(r0.copy(), dev)
    lo.mul(res, op, T(v, dev), 3.0, -4.0)
    assert rel(res.cpu().numpy(), 3.0 * (A @ v) - 4.0 * r0) <= tol
    fl = oracle.SCALARS_F64 if dtype == torch.float32 else 0
    assert rel(res.cpu().numpy(), oracle.gemv(r0.copy(), A, v, 3.0, -4.0, flags=fl)) <= tol
    assert lo.nprod(op) == 2 and lo.ntprod(op) == 1 and lo.nctprod(op) == 1
    with pytest.raises(lo.LinearOperatorException):
        op * torch.ones(n + 1, dtype=dtype, device=dev)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("n", [1, 10, 257, 1500])
def test_hermitian(lo, dev, dtype, n):
    """test_linop.jl:360-380."""
    rng = np.random.default_rng(n)
    npd = NP[dtype]
    A = rng.standard_normal((n, n)).astype(npd)
    d = rng.standard_normal(n).astype(npd)
    L = np.tril(A.astype(np.float64), -1)
    Cm = L + L.T + np.diag(d.astype(np.float64))
    v = SV(n).astype(npd)
    tol = 1e-12 if dtype == torch.float64 else 3e-5
    H = lo.opHermitian(T(d, dev), TM(A, dev))
    assert lo.ishermitian(H) and lo.issymmetric(H)
    for op in (H, H.T, H.H):
        assert rel((op * T(v, dev)).cpu().numpy(), Cm @ v) <= tol
    r0 = rng.standard_normal(n).astype(npd)
    res = T(r0.copy(), dev)
    lo.mul(res, H, T(v, dev), 3.0, -4.0)
    fl = oracle.SCALARS_F64 if dtype == torch.float32 else 0
    assert rel(res.cpu().numpy(), oracle.hermitian_mul(r0.copy(), d, A, v, 3.0, -4.0, flags=fl)) <= tol
    Csym = (A + A.T).astype(npd)
    H2 = lo.opHermitian(TM(Csym, dev))
    assert rel((H2 * T(v, dev)).cpu().numpy(), Csym.astype(np.float64) @ v) <= tol
    with pytest.raises(lo.LinearOperatorException):
        lo.opHermitian(T(d, dev)[: max(n - 1, 0)], TM(A, dev)) if n > 1 else (_ for _ in ()).throw(lo.LinearOperatorException("x"))


@pytest.mark.parametrize("n,dtype", [(5700, torch.float64), (5889, torch.float64), (5700, torch.float32),
                                     (11776, torch.float64), (11585, torch.float64), (11586, torch.float32),
                                     (12800, torch.float32), (12803, torch.float32)])
def test_hermitian_strip_regimes(lo, dev, dtype, n):
    """Every strip length of the single-pass kernel (1, 2, 8 tiles per workgroup), ragged last row groups and the
    unaligned (odd leading dimension) path, against the oracle restatement of mulHermitian! (linalg.jl:97-103)."""
    rng = np.random.default_rng(n)
    npd = NP[dtype]
    A = rng.standard_normal((n, n)).astype(npd)
    d = rng.standard_normal(n).astype(npd)
    v = SV(n).astype(npd)
    r0 = rng.standard_normal(n).astype(npd)
    tol = 1e-12 if dtype == torch.float64 else 3e-5
    H = lo.opHermitian(T(d, dev), TM(A, dev))
    fl = oracle.SCALARS_F64 if dtype == torch.float32 else 0
    res = T(r0.copy(), dev)
    lo.mul(res, H, T(v, dev), 3.0, -4.0)
    assert rel(res.cpu().numpy(), oracle.hermitian_mul(r0.copy(), d, A, v, 3.0, -4.0, flags=fl)) <= tol
    res2 = T(np.full(n, np.nan, dtype=npd), dev)                 # beta == 0 never reads res
    lo.mul(res2, H, T(v, dev), 1.0, 0.0)
    assert rel(res2.cpu().numpy(), oracle.hermitian_mul(r0.copy(), d, A, v, 1.0, 0.0, flags=fl)) <= tol
    res3 = T(np.full(n, np.nan, dtype=npd), dev)                 # deterministic: bit-identical on repeat
    lo.mul(res3, H, T(v, dev), 1.0, 0.0)
    assert torch.equal(res2, res3)


@pytest.mark.parametrize("n,dtype", [(7, torch.float64), (300, torch.float64), (1024, torch.float32), (2051, torch.float64),
                                     (5700, torch.float64), (5889, torch.float32), (11776, torch.float64), (4096, torch.float64)])
def test_hermitian_block_apply_is_bit_identical_to_the_column_loop(lo, dev, n, dtype):
    """Round 5: `mul!(res::Matrix, opHermitian(d, A), V::Matrix, α, β)` — the reference's closure applied
    to the columns of a matrix (src/linalg.jl:97-103 through src/operations.jl:34-36) — reads the strict lower triangle ONCE
    per chunk of up to 4 columns (mxlo_hermitian_mul_block). Per column the arithmetic and every addition order are those
    of the single apply: bit-identical to the column loop for k = 1 … 7 (chunks of 4, 2 and 1), every strip regime, ragged
    and unaligned shapes (odd leading dimensions of A, V and res), α / β forms, NaN above the diagonal; and against the
    oracle."""
    from linearoperators_jl_amd.device import get_ctx
    ctx = get_ctx(dev)
    npd = NP[dtype]
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n)).astype(npd)
    A[np.triu_indices(n)] = np.nan
    d = rng.standard_normal(n).astype(npd)
    tol = 1e-12 if dtype == torch.float64 else 3e-5
    fl = oracle.SCALARS_F64 if dtype == torch.float32 else 0

    def cm(X, pad):                                                # column-major device matrix, leading dimension n + pad
        big = torch.zeros(X.shape[1], X.shape[0] + pad, dtype=dtype, device=dev)
        big[:, :X.shape[0]] = T(np.ascontiguousarray(X.T), dev)
        return big[:, :X.shape[0]].t()

    ctx.tune("herm_single", 0)                                     # the column loop below must be the same two-launch form
    try:
        for k, pad, (a, b) in ((1, 0, (1.0, 0.0)), (2, 1, (2.0, -3.0)), (3, 0, (1.0, 0.0)), (4, 2, (1.5, 0.5)), (7, 1, (2.0, -3.0))):
            H = lo.opHermitian(T(d, dev), cm(A, pad % 2))
            Vh, R0 = rng.standard_normal((n, k)).astype(npd), rng.standard_normal((n, k)).astype(npd)
            Vd = cm(Vh, pad)
            res = cm(R0, pad)
            lo.mul(res, H, Vd, a, b)                               # the block entry point
            cols = cm(R0, 0)
            for j in range(k):
                lo.mul(cols[:, j], H, Vd[:, j].contiguous(), a, b)
            assert torch.equal(res, cols), (n, k, pad)
            want = np.stack([oracle.hermitian_mul(R0[:, j].copy(), d, np.tril(A, -1), Vh[:, j].copy(), a, b, flags=fl) for j in range(k)], axis=1)
            assert rel(res.cpu().numpy(), want) <= tol, (n, k)
    finally:
        ctx.tune("herm_single", 1)
        ctx.tune("kron_fuse", 1)
    with pytest.raises(lo.LinearOperatorException):
        lo.mul(torch.empty(n, 2, dtype=dtype, device=dev), H, torch.empty(n, 3, dtype=dtype, device=dev), 1.0, 0.0)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("n", [2304, 4096, 5000])
def test_hermitian_strip_order_does_not_change_a_bit(lo, dev, dtype, n):
    """Round 6: the interior strips of opHermitian are walked column block by column block (tune key herm_order = 1, the
    default: workgroups that run together read contiguous runs down the same columns — n = 4096 f64 17.7 -> 16.2 us) instead
    of row group by row group (0). Every partial lands in the same slot and is added in the same order: single applies and
    block applies are bit-identical under both orders."""
    g = torch.Generator(device="cpu").manual_seed(n)
    M = (torch.rand(n, n, dtype=dtype, generator=g) - 0.5).to(dev).t()
    d, x = ((torch.rand(n, dtype=dtype, generator=g) - 0.5).to(dev) for _ in range(2))
    V = (torch.rand(4, n, dtype=dtype, generator=g) - 0.5).to(dev).t()
    H = lo.opHermitian(d, M)
    ctx = lo.get_ctx(dev)
    got = {}
    try:
        for order in (0, 1):
            ctx.tune("herm_order", order)
            r = torch.empty(n, dtype=dtype, device=dev)
            lo.mul(r, H, x, 0.7, 0.0)
            R = torch.empty(4, n, dtype=dtype, device=dev).t()
            lo.mul(R, H, V, 1.0, 0.0)
            got[order] = (r, R)
    finally:
        ctx.tune("herm_order", 1)
    assert torch.equal(got[0][0], got[1][0]) and torch.equal(got[0][1], got[1][1])


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32, torch.complex128])
@pytest.mark.parametrize("n", [1000, 2048, 4096])
def test_hermitian_load_policy_poll_interval_and_strip_width_do_not_change_the_result(lo, dev, dtype, n):
    """Round 6, second half: the strip loads carry the nontemporal hint or not by triangle size (tune key herm_nt — a TEMPLATE
    parameter of the pass kernels: under a run-time flag hipcc merged the two load sequences and dropped the hint), the
    finishers of the single launch poll every `herm_poll_sleep` x 64 clocks, and the single launch takes 2-tile strips from
    1.25 per CU on. The first two cannot change a bit (single and block applies); the strip width regroups the row partials:
    1e-12 / 3e-5 against the default."""
    g = torch.Generator(device="cpu").manual_seed(n + 7)
    rdt = {torch.float64: torch.float64, torch.float32: torch.float32, torch.complex128: torch.float64}[dtype]
    mk = lambda *shape: ((torch.rand(*shape, dtype=rdt, generator=g) - 0.5) if not dtype.is_complex
                         else torch.complex(torch.rand(*shape, dtype=rdt, generator=g) - 0.5, torch.rand(*shape, dtype=rdt, generator=g) - 0.5))
    M = mk(n, n).to(dev).t()
    d = (torch.rand(n, dtype=rdt, generator=g) - 0.5).to(dev)
    x = mk(n).to(dev)
    H = lo.opHermitian(d, M)
    ctx = lo.get_ctx(dev)

    def run():
        r = torch.empty(n, dtype=dtype, device=dev)
        lo.mul(r, H, x, 0.7, 0.0)
        if dtype.is_complex:
            return r, None
        V = (torch.rand(3, n, dtype=dtype, generator=torch.Generator(device="cpu").manual_seed(1)) - 0.5).to(dev).t()
        R = torch.empty(3, n, dtype=dtype, device=dev).t()
        lo.mul(R, H, V, 1.0, 0.0)
        return r, R

    try:
        base = run()
        for key, values, default in (("herm_nt", (0, 1), -1), ("herm_poll_sleep", (1, 64), 4)):
            for val in values:
                ctx.tune(key, val)
                got = run()
                assert torch.equal(got[0], base[0]) and (base[1] is None or torch.equal(got[1], base[1])), (key, val)
            ctx.tune(key, default)
        if not dtype.is_complex:
            for width in (1, 2):
                ctx.tune("herm_strip", width)
                got = run()
                tol = 1e-12 if dtype == torch.float64 else 3e-5
                assert float((got[0].double() - base[0].double()).norm() / base[0].double().norm()) <= tol, width
            ctx.tune("herm_strip", 0)
    finally:
        ctx.tune("herm_nt", -1)
        ctx.tune("herm_poll_sleep", 4)
        ctx.tune("herm_strip", 0)


def test_hermitian_single_launch_is_bit_identical_to_the_two_launch_form(lo, dev):
    """Round 5: for full row groups of an aligned matrix (n a multiple of 256 / 512, n <= 8192)
    opHermitian is ONE launch — strip workgroups publish their partials as self-validating slots, finisher workgroups of
    the same launch wait for exactly the slots their rows need, add them in the order of the separate finish launch and
    re-arm them. Bit-identical to the two-launch form (`herm_single` = 0) and to itself from run to run, across changes of
    n, element type and strip shape between applies (the slot layout changes: re-arm), NaN above the diagonal, alpha / beta
    forms, 200 back-to-back applies — and against the oracle (src/linalg.jl:97-103)."""
    from linearoperators_jl_amd.device import get_ctx
    ctx = get_ctx(dev)
    rng = np.random.default_rng(4)
    cases = [(torch.float64, 256), (torch.float64, 4096), (torch.float32, 4096), (torch.float64, 1024), (torch.float64, 8192),
             (torch.float32, 1024), (torch.float64, 4096), (torch.float64, 2048 + 256)]
    ops = {}

    def launches():
        import ctypes as C
        a = (C.c_int64 * 12)()
        lo._lib.call("mxlo_debug_counters", a)
        return a[10]

    ctx.tune("herm_single_max_n", 8192)                           # (the default rule is by size — 112 MiB of triangle —, see common.h)
    for dtype, n in cases:
        npd = NP[dtype]
        if (dtype, n) not in ops:
            A = rng.standard_normal((n, n)).astype(npd)
            A[np.triu_indices(n)] = np.nan                        # never read
            d, v, r0 = (rng.standard_normal(n).astype(npd) for _ in range(3))
            ops[(dtype, n)] = (lo.opHermitian(T(d, dev), TM(A, dev)), A, d, v, r0)
        H, A, d, v, r0 = ops[(dtype, n)]
        got = {}
        for single in (1, 0, 1):
            ctx.tune("herm_single", single)
            try:
                res = T(r0.copy(), dev)
                l0 = launches()
                lo.mul(res, H, T(v, dev), 3.0, -4.0)
                nl = launches() - l0
                assert (nl <= 2) if single else (nl == 2), (dtype, n, single, nl)   # 1, or 2 when the slots were re-armed for a new layout
                got.setdefault(single, []).append(res.cpu().numpy())
            finally:
                ctx.tune("herm_single", 1)
                ctx.tune("kron_fuse", 1)
        assert np.array_equal(got[1][0], got[0][0]) and np.array_equal(got[1][0], got[1][1]), (dtype, n)
        fl = oracle.SCALARS_F64 if dtype == torch.float32 else 0
        want = oracle.hermitian_mul(r0.copy(), d, np.tril(A, -1), v, 3.0, -4.0, flags=fl)
        assert rel(got[1][0], want) <= (1e-12 if dtype == torch.float64 else 3e-5), (dtype, n)
    H, A, d, v, r0 = ops[(torch.float64, 4096)]
    vt, res = T(v, dev), torch.empty(4096, dtype=torch.float64, device=dev)
    lo.mul(res, H, vt, 1.0, 0.0)
    first = res.clone()
    l0 = launches()
    for _ in range(200):
        lo.mul(res, H, vt, 1.0, 0.0)
    assert launches() - l0 == 200                                  # ONE launch per apply
    assert torch.equal(res, first)
    ctx.tune("herm_single_max_n", 0)                               # back to the by-size rule


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32, torch.complex128, torch.complex64])
@pytest.mark.parametrize("n", [7, 256, 300, 1024, 2051])
def test_hermitian_reads_strict_lower_triangle_only(lo, dev, dtype, n):
    """opHermitian keeps tril(A, -1) (src/linalg.jl:114): whatever the caller's A holds on and above the diagonal — NaN
    here — must not reach the result, and the result must equal the one of the same A with zeros up there, bit for bit.
    (The diagonal-block tiles of full row groups load unmasked and SELECT the strict lower triangle.)"""
    rng = np.random.default_rng(n)
    cplx = dtype.is_complex
    npd = {torch.float64: np.float64, torch.float32: np.float32, torch.complex128: np.complex128,
           torch.complex64: np.complex64}[dtype]
    rd = lambda *sh: (rng.uniform(-1, 1, sh) + (1j * rng.uniform(-1, 1, sh) if cplx else 0)).astype(npd)
    A = rd(n, n)
    clean, dirty = np.tril(A, -1), A.copy()
    dirty[np.triu_indices(n)] = np.nan
    d, v = rd(n), rd(n)
    outs = []
    for M in (clean, dirty):
        H = lo.opHermitian(T(d, dev), TM(M, dev))
        res = torch.empty(n, dtype=dtype, device=dev)
        lo.mul(res, H, T(v, dev), 1.0, 0.0)
        outs.append(res)
    assert torch.isfinite(torch.view_as_real(outs[1]) if cplx else outs[1]).all()
    assert torch.equal(outs[0], outs[1])
    want = d.astype(np.complex128) * v + clean.astype(np.complex128) @ v + clean.astype(np.complex128).conj().T @ v
    got = outs[0].cpu().numpy().astype(np.complex128)
    assert np.linalg.norm(got - want) <= (1e-12 if npd in (np.float64, np.complex128) else 3e-5) * np.linalg.norm(want)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("shapes", [((3, 5), (4, 2)), ((1, 1), (7, 3)), ((16, 16), (16, 16)), ((70, 33), (65, 129)),
                                    ((64, 64), (128, 64))])
def test_kron(lo, dev, dtype, shapes):
    """test_kron.jl:2-39: K = kron(A,B) dense vs the operator, its transpose and adjoint, 5-arg form."""
    (m, n), (p, q) = shapes
    rng = np.random.default_rng(m + 10 * n + 100 * p + 1000 * q)
    npd = NP[dtype]
    A, B = rng.standard_normal((m, n)).astype(npd), rng.standard_normal((p, q)).astype(npd)
    K = np.kron(A.astype(np.float64), B.astype(np.float64))
    Kop = lo.kron(lo.LinearOperatorFromMatrix(TM(A, dev)), TM(B, dev))
    assert Kop.shape == K.shape
    x, xt = rng.standard_normal(K.shape[1]).astype(npd), rng.standard_normal(K.shape[0]).astype(npd)
    tol = 1e-12 if dtype == torch.float64 else 3e-5
    n1 = np.linalg.norm(K, 1)
    assert np.linalg.norm((Kop * T(x, dev)).cpu().numpy() - K @ x, 1) <= tol * n1 * max(1, np.abs(x).max())
    assert rel((Kop.T * T(xt, dev)).cpu().numpy(), K.T @ xt) <= tol * 10
    assert rel((Kop.H * T(xt, dev)).cpu().numpy(), K.T @ xt) <= tol * 10
    r0 = rng.standard_normal(K.shape[0]).astype(npd)
    res = T(r0.copy(), dev)
    lo.mul(res, Kop, T(x, dev), 2.0, 3.0)
    fl = oracle.SCALARS_F64 if dtype == torch.float32 else 0
    assert rel(res.cpu().numpy(), oracle.kron_mul(r0.copy(), A, B, x, 2.0, 3.0, flags=fl)) <= tol * 10
    res = T(np.full(K.shape[0], np.nan, dtype=npd), dev)
    lo.mul(res, Kop, T(x, dev), 1.0, 0.0)                  # beta == 0 never reads res
    assert torch.isfinite(res).all()


def test_kron_transpose_detecting_and_scaling(lo, dev):
    """Asymmetric identity check of the MFMA C/D layout + test_kron.jl:50-58 (2*kron(opEye(2), I(1)))."""
    n = 48
    A = np.eye(n)
    B = np.arange(n * n, dtype=np.float64).reshape(n, n) / n      # asymmetric
    Kop = lo.kron(TM(A, dev), TM(B, dev))
    x = np.random.default_rng(0).standard_normal(n * n)
    assert rel((Kop * T(x, dev)).cpu().numpy(), np.kron(A, B) @ x) <= 1e-13
    K2 = 2 * lo.kron(lo.opEye(2), torch.eye(1, dtype=torch.float64, device=dev))
    assert np.allclose(lo.Matrix(K2).cpu().numpy(), 2 * np.eye(2))


def test_kron_1024_properties(lo, dev):
    """BASELINE config 4b size: (A⊗B)x = vec(B X Aᵀ) against torch's own fp64 GEMMs."""
    g = torch.Generator(device=dev).manual_seed(3)
    n = 1024
    A = ((torch.rand(n, n, dtype=torch.float64, device=dev, generator=g) * 2 - 1) / 32).t()
    B = ((torch.rand(n, n, dtype=torch.float64, device=dev, generator=g) * 2 - 1) / 32).t()
    x = torch.rand(n * n, dtype=torch.float64, device=dev, generator=g) * 2 - 1
    Kop = lo.kron(A, B)
    got = Kop * x
    X = x.view(n, n).t()                                   # reshape(x, q, n) column-major
    want = (B @ X @ A.t()).t().reshape(-1)
    assert (torch.linalg.vector_norm(got - want) / torch.linalg.vector_norm(want)).item() <= 1e-12
    gt = Kop.T * x
    Xt = x.view(n, n).t()
    want = (B.t() @ Xt @ A).t().reshape(-1)
    assert (torch.linalg.vector_norm(gt - want) / torch.linalg.vector_norm(want)).item() <= 1e-12


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_blockdiag_fused(lo, dev, dtype):
    """test_linop.jl:718-756: mixes operators and plain matrices; Matrix(M), transpose, adjoint vs dense."""
    rng = np.random.default_rng(2)
    npd = NP[dtype]
    S = lo.Storage(dtype, dev)
    d1, d2 = rng.standard_normal(5).astype(npd), rng.standard_normal(1031).astype(npd)
    M1, M2 = rng.standard_normal((4, 7)).astype(npd), rng.standard_normal((300, 3)).astype(npd)
    ops = [lo.opDiagonal(T(d1, dev)), TM(M1, dev), lo.opEye(dtype, 3, S=S), lo.LinearOperatorFromMatrix(TM(M2, dev)),
           lo.opZeros(dtype, 2, 5, S=S), lo.opDiagonal(T(d2, dev))]
    dense = [np.diag(d1), M1, np.eye(3), M2, np.zeros((2, 5)), np.diag(d2)]
    nr, nc = sum(a.shape[0] for a in dense), sum(a.shape[1] for a in dense)
    D = np.zeros((nr, nc))
    r = c = 0
    for a in dense:
        D[r:r + a.shape[0], c:c + a.shape[1]] = a
        r += a.shape[0]; c += a.shape[1]
    BD = lo.BlockDiagonalOperator(*ops)
    assert hasattr(BD, "_keepalive")                       # took the single-launch path
    tol = 1e-12 if dtype == torch.float64 else 2e-5
    x, xt = rng.standard_normal(nc).astype(npd), rng.standard_normal(nr).astype(npd)
    assert rel((BD * T(x, dev)).cpu().numpy(), D @ x) <= tol
    assert rel((BD.T * T(xt, dev)).cpu().numpy(), D.T @ xt) <= tol
    assert rel((BD.H * T(xt, dev)).cpu().numpy(), D.T @ xt) <= tol
    r0 = rng.standard_normal(nr).astype(npd)
    res = T(r0.copy(), dev)
    lo.mul(res, BD, T(x, dev), 2.0, -3.0)
    assert rel(res.cpu().numpy(), 2.0 * (D @ x) - 3.0 * r0) <= tol
    res = T(np.full(nr, np.nan, dtype=npd), dev)
    lo.mul(res, BD, T(x, dev), 1.0, 0.0)
    assert torch.isfinite(res).all()


def test_blockdiag_of_diagonals_bit_exact_and_generic_path(lo, dev):
    """BASELINE config 4a shape (1024 opDiagonal blocks) at a reduced block size with ODD block length so
    blocks start at every 16-byte phase: the fused launch must equal block-by-block oracle results bit
    for bit; the generic path (a Householder block forces the reference's loop structure) agrees too."""
    rng = np.random.default_rng(5)
    nb, bs = 1024, 997
    ds = [rng.standard_normal(bs) for _ in range(nb)]
    BD = lo.BlockDiagonalOperator(*[lo.opDiagonal(T(d, dev)) for d in ds])
    x, r0 = rng.standard_normal(nb * bs), rng.standard_normal(nb * bs)
    for alpha, beta in ((1.0, 0.0), (2.0 / 3.0, -0.3)):
        res = T(r0.copy(), dev)
        lo.mul(res, BD, T(x, dev), alpha, beta)
        want = r0.copy()
        for k, d in enumerate(ds):
            sl = slice(k * bs, (k + 1) * bs)
            want[sl] = oracle.diag_mul(r0[sl].copy(), d, np.ascontiguousarray(x[sl]), alpha, beta)
        assert np.array_equal(res.cpu().numpy(), want)
    h = rng.standard_normal(50); h /= np.linalg.norm(h)
    G = lo.BlockDiagonalOperator(lo.opDiagonal(T(ds[0], dev)), lo.opHouseholder(T(h, dev)), lo.opDiagonal(T(ds[1], dev)))
    assert not hasattr(G, "_keepalive")
    xx = rng.standard_normal(2 * bs + 50)
    want = np.concatenate([ds[0] * xx[:bs], xx[bs:bs + 50] - 2 * (h @ xx[bs:bs + 50]) * h, ds[1] * xx[bs + 50:]])
    assert rel((G * T(xx, dev)).cpu().numpy(), want) <= 1e-12
    assert rel((G.T * T(xx, dev)).cpu().numpy(), want) <= 1e-12


def test_combinators_vs_dense(lo, dev):
    """test_linop.jl (arithmetic), test_cat.jl:96-118,165-187 (α,β = 3,-4), test_adjtrans.jl."""
    rng = np.random.default_rng(9)
    n = 40
    A1, A2 = rng.standard_normal((n, n)), rng.standard_normal((n, n))
    d = rng.standard_normal(n)
    h = rng.standard_normal(n); h /= np.linalg.norm(h)
    opA1, opA2 = lo.LinearOperatorFromMatrix(TM(A1, dev)), lo.LinearOperatorFromMatrix(TM(A2, dev))
    D, H = lo.opDiagonal(T(d, dev)), lo.opHouseholder(T(h, dev))
    Hd = np.eye(n) - 2 * np.outer(h, h)
    cases = [
        (opA1 + opA2, A1 + A2), (opA1 - opA2, A1 - A2), (-opA1, -A1), (opA1 * opA2, A1 @ A2),
        (2.5 * opA1, 2.5 * A1), (opA1 * 2.5, 2.5 * A1), (opA1 / 4.0, A1 / 4.0),
        (H * D * H.H, Hd @ np.diag(d) @ Hd.T), (opA1 + 1.5, A1 + 1.5), (2.0 - opA1, 2.0 - A1),
        ((opA1 * D + H).T, (A1 @ np.diag(d) + Hd).T), ((opA1 * opA2).H, (A1 @ A2).T),
        (lo.hcat(opA1, D, H), np.hstack([A1, np.diag(d), Hd])), (lo.vcat(opA1, D, H), np.vstack([A1, np.diag(d), Hd])),
        (lo.hvcat((2, 2), opA1, D, H, opA2), np.block([[A1, np.diag(d)], [Hd, A2]])),
        (lo.hcat(opA1, D).T, np.hstack([A1, np.diag(d)]).T), (lo.vcat(opA1, D).H, np.vstack([A1, np.diag(d)]).T),
    ]
    # op[rows, cols] = R * op * E (special-operators.jl:225-233). mulRestrict!/multRestrict! IGNORE α and β
    # (:167-174), so a 5-arg mul! through an outer restriction returns the plain product — reference
    # behaviour, reproduced here.
    index_cases = [
        (opA1[[3, 4], [5, 6]], A1[np.ix_([2, 3], [4, 5])]),            # test_linop.jl:463-466
        (opA1[lo.jrange(2, 9, 3), slice(None)], A1[1:9:3, :]),
        (opA1[4, lo.jrange(1, n)], A1[3:4, :]),
    ]
    for op, M in cases + index_cases:
        m, k = M.shape
        assert op.shape == (m, k)
        v, r0 = rng.standard_normal(k), rng.standard_normal(m)
        assert rel((op * T(v, dev)).cpu().numpy(), M @ v) <= 1e-12
        res = T(r0.copy(), dev)
        lo.mul(res, op, T(v, dev), 3.0, -4.0)
        if any(op is ic[0] for ic in index_cases):
            assert rel(res.cpu().numpy(), M @ v) <= 1e-12
        else:
            assert rel(res.cpu().numpy(), 3.0 * (M @ v) - 4.0 * r0) <= 1e-12
        assert rel(lo.Matrix(op).cpu().numpy(), M) <= 1e-12
    with pytest.raises(lo.LinearOperatorException):
        opA1 * lo.opDiagonal(T(d[:5], dev))
    with pytest.raises(lo.LinearOperatorException):
        opA1 + lo.opDiagonal(T(d[:5], dev))
    with pytest.raises(lo.LinearOperatorException):
        lo.hcat(opA1, lo.opDiagonal(T(d[:5], dev)))
    # wrapper identities (test_adjtrans.jl)
    assert lo.adjoint(lo.adjoint(opA1)) is opA1 and lo.transpose(lo.transpose(opA1)) is opA1
    assert isinstance(lo.transpose(lo.adjoint(opA1)), lo.ConjugateLinearOperator)
    assert lo.storage_type(opA1.T) == lo.storage_type(opA1)
    # counters through wrappers (test_linop.jl:634-716)
    lo.reset(opA1)
    v = T(rng.standard_normal(n), dev)
    opA1 * v; opA1.T * v; opA1.H * v; opA1.H * v
    assert (lo.nprod(opA1), lo.ntprod(opA1), lo.nctprod(opA1)) == (1, 1, 2)
    assert lo.nprod(opA1.H) == 2 and lo.nctprod(opA1.T) == 1


def test_kron_abi_transpose_mode(lo, dev):
    """mxlo_kron_mul with op_mode = T/C directly (the host glue normally feeds pre-transposed factors to
    the N mode): generic-transposition GEMM path, odd sizes exercising the tile guards."""
    from linearoperators_jl_amd import _lib
    from linearoperators_jl_amd.device import get_ctx, ptr
    rng = np.random.default_rng(12)
    for (m, n), (p, q) in (((5, 3), (4, 7)), ((64, 96), (128, 64)), ((70, 33), (65, 129))):
        A, B = rng.standard_normal((m, n)), rng.standard_normal((p, q))
        K = np.kron(A, B)
        Ad, Bd = TM(A, dev), TM(B, dev)
        x = rng.standard_normal(m * p)
        r0 = rng.standard_normal(n * q)
        res = T(r0.copy(), dev)
        work = torch.empty(max(q * m, p * n), dtype=torch.float64, device=dev)
        ctx = get_ctx(dev)
        for mode in (_lib.OP_T, _lib.OP_C):
            res.copy_(T(r0, dev))
            _lib.call("mxlo_kron_mul", ctx.handle, _lib.F64, ptr(res), ptr(Ad), m, n, Ad.stride(1), ptr(Bd), p, q,
                      Bd.stride(1), ptr(T(x, dev)), ptr(work), 2.0, 3.0, mode, 0)
            assert rel(res.cpu().numpy(), 2.0 * (K.T @ x) + 3.0 * r0) <= 1e-12
            assert rel(res.cpu().numpy(), oracle.kron_mul(r0.copy(), A, B, x, 2.0, 3.0, trans=True)) <= 1e-12


def test_shifted_operator(lo, dev):
    """src/shifted_operators.jl: (H + σI) x through the inner operator's mul! + one axpy; σ is mutable;
    works over leaves, quasi-Newton operators and composites; transpose/adjoint; counters; reset!."""
    rng = np.random.default_rng(21)
    n = 300
    A = rng.standard_normal((n, n))
    opA = lo.LinearOperatorFromMatrix(TM(A, dev))
    Sh = lo.ShiftedOperator(opA, 0.75)
    x, r0 = rng.standard_normal(n), rng.standard_normal(n)
    assert rel((Sh * T(x, dev)).cpu().numpy(), A @ x + 0.75 * x) <= 1e-13
    assert rel((Sh.T * T(x, dev)).cpu().numpy(), A.T @ x + 0.75 * x) <= 1e-13
    assert rel((Sh.H * T(x, dev)).cpu().numpy(), A.T @ x + 0.75 * x) <= 1e-13
    res = T(r0.copy(), dev)
    lo.mul(res, Sh, T(x, dev), 2.0, -3.0)
    assert rel(res.cpu().numpy(), 2.0 * (A @ x + 0.75 * x) - 3.0 * r0) <= 1e-13
    Sh.data.sigma = 0.0                                    # σ == 0: the axpy is skipped (:21)
    assert rel((Sh * T(x, dev)).cpu().numpy(), A @ x) <= 1e-13
    assert not lo.issymmetric(Sh) and lo.has_args5(Sh) and lo.isallocated5(Sh)
    assert lo.nprod(Sh) == 3 and lo.ntprod(Sh) == 1 and lo.nctprod(Sh) == 1
    lo.reset(Sh)
    assert lo.nprod(Sh) == 0
    with pytest.raises(ValueError):
        lo.ShiftedOperator(lo.LinearOperatorFromMatrix(TM(A[:, :5], dev)), 1.0)
    # over a forward L-BFGS operator: (B + σI) p = b is what solve_shifted_system! inverts
    B = lo.LBFGSOperator(n, mem=5, device=dev)
    for _ in range(7):
        s = rng.uniform(-1, 1, n)
        y = s * rng.uniform(0.5, 2.0, n)
        lo.push(B, T(s, dev), T(y, dev))
    Bs = lo.ShiftedOperator(B, 0.3)
    assert lo.issymmetric(Bs) and lo.ishermitian(Bs)
    p = lo.solve_shifted_system(torch.zeros(n, dtype=torch.float64, device=dev), B, Bs * T(x, dev), 0.3)
    assert rel(p.cpu().numpy(), x) <= 1e-9


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-12), (torch.float32, 3e-5)])
@pytest.mark.parametrize("m,n", [(1000, 777), (513, 2050), (64, 3), (2, 5000), (4099, 133), (2050, 64), (8192, 31), (8200, 300),
                                 (4096, 1030), (8192, 2051), (16384, 1024), (1100, 8192)])   # round 5: tall enough for the LDS-staged transposed form; the last four: sizes at which the N mode takes its row-band kernels
@pytest.mark.parametrize("k", [2, 3, 5, 8, 11])
def test_block_gemv_vs_columns_and_dense(lo, dev, dtype, tol, m, n, k):
    """mul!(res::Matrix, LinearOperator(M), V::Matrix, α, β) through mxlo_gemv_block (M read once per 8 columns) against
    NumPy and against the column-by-column GEMVs, for M·V, transpose(M)·U and the 5-arg form, column-major and row-major M,
    aligned and odd leading dimensions of the operands."""
    npd = NP[dtype]
    rng = np.random.default_rng(m + 7 * n + 13 * k)
    Mh = rng.uniform(-1, 1, (m, n)).astype(npd)
    Vh, Uh = rng.uniform(-1, 1, (n, k)).astype(npd), rng.uniform(-1, 1, (m, k)).astype(npd)
    r0, r1 = rng.uniform(-1, 1, (m, k)).astype(npd), rng.uniform(-1, 1, (n, k)).astype(npd)
    W = Mh.astype(np.float64)

    def cm(X, pad=0):                                              # column-major device matrix, leading dimension + pad
        big = torch.zeros(X.shape[1], X.shape[0] + pad, dtype=dtype, device=dev)
        big[:, :X.shape[0]] = T(np.ascontiguousarray(X.T), dev)
        return big[:, :X.shape[0]].t()

    for Md in (TM(Mh, dev), T(Mh, dev)):
        op = lo.LinearOperatorFromMatrix(Md)
        for pad in (0, 1):
            res = cm(r0, pad)
            lo.mul(res, op, cm(Vh, pad), 2.0, -0.5)
            assert rel(res.cpu().numpy(), 2.0 * (W @ Vh) - 0.5 * r0) <= tol
            rt = cm(r1, pad)
            lo.mul(rt, op.T, cm(Uh, pad), 1.0, 0.0)
            assert rel(rt.cpu().numpy(), W.T @ Uh) <= tol
            cols = torch.empty(k, n, dtype=dtype, device=dev).t()
            for j in range(k):                                         # the same product, one GEMV per column
                lo.mul(cols[:, j], op.T, cm(Uh, pad)[:, j].contiguous(), 1.0, 0.0)
            assert rel(rt.cpu().numpy(), cols.cpu().numpy()) <= tol
    res = torch.full((m, k), float("nan"), dtype=dtype, device=dev).t().contiguous().t()
    lo.mul(res, op, cm(Vh))                                            # beta == 0 never reads res
    assert rel(res.cpu().numpy(), W @ Vh) <= tol


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-12), (torch.float32, 3e-5)])
@pytest.mark.parametrize("m,n", [(16384, 1024), (16392, 1500), (8192, 2051), (8200, 1030), (32768, 1027), (32772, 1155), (4096, 1100), (4100, 1024)])
@pytest.mark.parametrize("k", [2, 4, 5, 8, 11])
def test_block_gemv_row_bands_are_bit_identical_to_the_single_applies(lo, dev, dtype, tol, m, n, k):
    """Round 6: where `M·v` runs as row bands of 512 / 256 bytes per column, `M·V` on a block runs the same bands with V
    staged in LDS (gemvb_n_rows_kernel): one launch, no partial workspace. Column assignment, fma order and the fixed-order
    sum over the column lanes are the single apply's, so column c of the block has the BITS of `mul!(res[:, c], op, V[:, c])`
    (src/operations.jl:34-36 applied to the closure of src/constructors.jl:25-27) — ragged last band (m not a multiple of
    the band), columns past the last whole step, every (α, β) form; with the knob off the column-chunk schedule agrees to
    the tolerance."""
    npd = NP[dtype]
    rng = np.random.default_rng(m + 3 * n + k)
    Mh = rng.uniform(-1, 1, (m, n)).astype(npd)
    Vh = rng.uniform(-1, 1, (n, k)).astype(npd)
    r0 = rng.uniform(-1, 1, (m, k)).astype(npd)
    op = lo.LinearOperatorFromMatrix(TM(Mh, dev))
    V = TM(Vh, dev)
    ctx = lo.get_ctx(dev)
    vr, ncu = 16 // np.dtype(npd).itemsize, torch.cuda.get_device_properties(dev).multi_processor_count
    banded = m % vr == 0 and m >= 8 * vr * ncu and n < 16384   # both forms take the same row band (dense.hip gemv_rows_band)
    for alpha, beta in ((1.0, 0.0), (2.0, -0.5), (-1.5, 1.0)):
        res = TM(r0, dev)
        lo.mul(res, op, V, alpha, beta)
        cols = TM(r0, dev)
        for j in range(k):
            lo.mul(cols[:, j], op, V[:, j], alpha, beta)
        if banded:
            assert torch.equal(res, cols), (alpha, beta)
        ref = alpha * (Mh.astype(np.float64) @ Vh) + beta * r0
        assert rel(res.cpu().numpy(), ref) <= tol
    ctx.tune("gemvb_n_rows", 0)
    try:
        old = TM(r0, dev)
        lo.mul(old, op, V, 2.0, -0.5)
    finally:
        ctx.tune("gemvb_n_rows", 1)
    assert rel(old.cpu().numpy(), 2.0 * (Mh.astype(np.float64) @ Vh) - 0.5 * r0) <= tol


def test_type_specific_operator_testset(lo, dev):
    """test/test_linop.jl:552-586 "Type specific operator": closures written by the caller, eltype(op) == T and
    op * w == T[2; 1] with eltype(op * w) == T for every element type; Matrix(opC) of a ComplexF64 closure operator.
    The closures index device vectors; nothing but the host routing is involved."""
    def prod(res, v, a, b):
        res[0] = v[0] + v[1]
        res[1] = v[1]

    def ctprod(res, v, a, b):
        res[0] = v[0]
        res[1] = v[0] + v[1]

    for T in (torch.complex128, torch.complex64, torch.float64, torch.float32, torch.float16, torch.int32):
        op = lo.LinearOperator(T, 2, 2, False, False, prod, None, ctprod, S=lo.Storage(T, dev))
        w = torch.ones(2, dtype=T, device=dev)
        assert lo.eltype(op) == T
        y = op * w
        assert y.dtype == T and torch.equal(y, torch.tensor([2, 1], dtype=T, device=dev))
        z = op.H * w                                                   # ctprod! is used as it is
        assert torch.equal(z, torch.tensor([1, 2], dtype=T, device=dev))
    A = torch.tensor([[1j, 1.0], [0.0, 1.0]], dtype=torch.complex128, device=dev)
    opC = lo.LinearOperator(torch.complex128, 2, 2, False, False, lambda r, v, a, b: r.copy_(A @ v),
                            lambda r, u, a, b: r.copy_(A.t() @ u), lambda r, w, a, b: r.copy_(A.conj().t() @ w),
                            S=lo.Storage(torch.complex128, dev))
    assert torch.equal(lo.Matrix(opC), A)
    assert torch.equal(lo.Matrix(opC.T), A.t()) and torch.equal(lo.Matrix(opC.H), A.conj().t())


def test_callable_functor(lo, dev):
    """test/test_callable.jl:1-21: a functor as prod! — `Flip` computes res = (-α) x (+ β res)."""
    class Flip:
        def __call__(self, res, x, alpha, beta):
            lo.leaves.mulOpEye(res, x, -alpha, beta, res.numel())

    S = lo.Storage(torch.float64, dev)
    op = lo.LinearOperator(torch.float64, 2, 2, True, True, Flip(), None, None, S=S)
    one = torch.ones(2, dtype=torch.float64, device=dev)
    assert torch.equal(op * one, -one) and torch.equal(op.H * one, -one) and torch.equal(op.T * one, -one)
    Mv = torch.ones(2, dtype=torch.float64, device=dev)
    lo.mul(Mv, op, one)
    assert torch.equal(Mv, -one) and lo.has_args5(op)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_kron_of_diagonal_factors_fused(lo, dev, dtype):
    """kron of opDiagonal / opEye factors takes the fused index-decomposition kernel: bit-exact against the
    reference's elementwise order dB[r]*(x*dA[c]) and equal to the dense Kronecker product (test_kron.jl:50-58)."""
    rng = np.random.default_rng(31)
    npd = NP[dtype]
    m, p = 37, 1031
    dA, dB = rng.standard_normal(m).astype(npd), rng.standard_normal(p).astype(npd)
    x, r0 = rng.standard_normal(m * p).astype(npd), rng.standard_normal(m * p).astype(npd)
    S = lo.Storage(dtype, dev)
    for A, B, a_, b_ in ((lo.opDiagonal(T(dA, dev)), lo.opDiagonal(T(dB, dev)), dA, dB),
                         (lo.opEye(dtype, m, S=S), lo.opDiagonal(T(dB, dev)), np.ones(m, npd), dB),
                         (lo.opDiagonal(T(dA, dev)), lo.opEye(dtype, p, S=S), dA, np.ones(p, npd))):
        K = lo.kron(A, B)
        assert K.shape == (m * p, m * p) and lo.issymmetric(K)
        want_inner = (b_[None, :] * (x.reshape(m, p) * a_[:, None])).reshape(-1)      # dB[r]*(x[r,c]*dA[c])
        for op in (K, K.T, K.H):
            res = torch.full((m * p,), float("nan"), dtype=dtype, device=dev)
            lo.mul(res, op, T(x, dev), 1, 0)
            assert np.array_equal(res.cpu().numpy(), want_inner)
        res = T(r0.copy(), dev)
        lo.mul(res, K, T(x, dev), np.float32(2) if dtype == torch.float32 else 2.0, np.float32(-3) if dtype == torch.float32 else -3.0)
        assert np.array_equal(res.cpu().numpy(), npd(2) * want_inner + npd(-3) * r0)
        assert rel((K * T(x, dev)).cpu().numpy(), np.kron(np.diag(a_.astype(np.float64)), np.diag(b_.astype(np.float64))) @ x) <= (1e-14 if dtype == torch.float64 else 1e-6)


def test_blockdiag_full_config_size(lo, dev):
    """BASELINE config 4a at its HBM size: 1024 opDiagonal blocks x 97,657 rows (odd length: every block starts
    at a different 16-byte phase) in ONE launch must equal, bit for bit, a single opDiagonal over the
    concatenated diagonal (same elementwise arithmetic), for beta == 0 and beta != 0, and its transpose."""
    nb, bs = 1024, 97_657
    g = torch.Generator(device=dev).manual_seed(8)
    dall = torch.rand(nb * bs, dtype=torch.float64, device=dev, generator=g) + 0.5
    x = torch.rand(nb * bs, dtype=torch.float64, device=dev, generator=g) * 2 - 1
    r0 = torch.rand(nb * bs, dtype=torch.float64, device=dev, generator=g)
    BD = lo.BlockDiagonalOperator(*[lo.opDiagonal(dall[k * bs:(k + 1) * bs]) for k in range(nb)])
    D = lo.opDiagonal(dall)
    assert hasattr(BD, "_keepalive") and BD.shape == (nb * bs, nb * bs)
    for alpha, beta in ((1.0, 0.0), (2.0 / 3.0, -0.3)):
        a, b = r0.clone(), r0.clone()
        lo.mul(a, BD, x, alpha, beta)
        lo.mul(b, D, x, alpha, beta)
        assert torch.equal(a, b)
        a.copy_(r0)
        lo.mul(a, BD.T, x, alpha, beta)
        assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("pad,off", [(6, (2, 4)), (5, (1, 3)), (8, (0, 0))])
def test_dense_views_with_leading_dimension(lo, dev, dtype, pad, off):
    """Column-major SubArray-style views: leading dimension > number of rows, base pointer at any element offset
    (aligned and unaligned 16-byte phases) — dense GEMV N/T, opHermitian, kron, BlockDiagonalOperator dense blocks."""
    npd = NP[dtype]
    rng = np.random.default_rng(pad)
    n, m2 = 301, 64
    tol = 1e-12 if dtype == torch.float64 else 3e-5
    big = rng.standard_normal((n + pad, n + pad)).astype(npd)
    Bdev = TM(big, dev)                                            # column-major (n+pad) x (n+pad)
    r0, c0 = off
    Av = Bdev[r0:r0 + n, c0:c0 + n]                                # view: ld = n + pad, offset pointer
    A = big[r0:r0 + n, c0:c0 + n]
    assert Av.stride(0) == 1 and Av.stride(1) == n + pad
    v, w = rng.standard_normal(n).astype(npd), rng.standard_normal(n).astype(npd)
    op = lo.LinearOperatorFromMatrix(Av)
    assert rel((op * T(v, dev)).cpu().numpy(), A.astype(np.float64) @ v) <= tol
    assert rel((op.T * T(w, dev)).cpu().numpy(), A.astype(np.float64).T @ w) <= tol
    d = rng.standard_normal(n).astype(npd)
    L = np.tril(A.astype(np.float64), -1)
    H = lo.opHermitian(T(d, dev), Av)
    assert rel((H * T(v, dev)).cpu().numpy(), (L + L.T + np.diag(d.astype(np.float64))) @ v) <= tol
    Kv = Bdev[r0:r0 + m2, c0:c0 + m2]
    K = lo.kron(Kv, Kv)
    xk = rng.standard_normal(m2 * m2).astype(npd)
    Kd = np.kron(big[r0:r0 + m2, c0:c0 + m2].astype(np.float64), big[r0:r0 + m2, c0:c0 + m2].astype(np.float64))
    assert np.linalg.norm((K * T(xk, dev)).cpu().numpy() - Kd @ xk) <= (1e-12 if dtype == torch.float64 else 3e-5) * np.linalg.norm(Kd, 1) * np.linalg.norm(xk)
    Bd = lo.BlockDiagonalOperator(Av, lo.opDiagonal(T(d, dev)))
    vv = np.concatenate([v, w])
    want = np.concatenate([A.astype(np.float64) @ v, d.astype(np.float64) * w])
    assert rel((Bd * T(vv, dev)).cpu().numpy(), want) <= tol


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("m,n,pad", [(2048, 1024, 0), (4096, 1030, 0), (4100, 2049, 4), (8192, 1024, 8), (16384, 1500, 0), (33000, 1024, 0)])
def test_dense_mul_row_band_kernel(lo, dev, dtype, m, n, pad):
    """Dense `M*v` (src/constructors.jl:19-29, N mode) through the single-launch row-band kernel (dense.hip:
    gemv_n_rows_kernel; m >= 2048 / 4096 rows, n >= 1024 columns, 16-byte aligned columns) against the dense model and
    against the two-launch column-chunk schedule it replaces (tune gemv_n_rows = 0): same tolerance class, 5-arg forms,
    NaN in `res` with beta = 0 never read, leading dimension > m, ragged last band."""
    from linearoperators_jl_amd.device import get_ctx
    ctx = get_ctx(dev)
    npd = NP[dtype]
    rng = np.random.default_rng(m + n)
    big = rng.standard_normal((m + pad, n)).astype(npd)
    Bdev = TM(big, dev)
    Av, A = Bdev[:m, :], big[:m, :].astype(np.float64)
    assert Av.stride(0) == 1 and Av.stride(1) == m + pad
    op = lo.LinearOperatorFromMatrix(Av)
    v = rng.standard_normal(n).astype(npd)
    r0 = rng.standard_normal(m).astype(npd)
    tol = 1e-12 if dtype == torch.float64 else 3e-5
    outs = {}
    for rows in (1, 0):
        ctx.tune("gemv_n_rows", rows)
        try:
            res = torch.full((m,), float("nan"), dtype=dtype, device=dev)
            lo.mul(res, op, T(v, dev), 1.0, 0.0)
            a = res.cpu().numpy()
            res2 = T(r0.copy(), dev)
            lo.mul(res2, op, T(v, dev), 3.0, -4.0)
            outs[rows] = (a, res2.cpu().numpy())
        finally:
            ctx.tune("gemv_n_rows", 1)
    want = A @ v.astype(np.float64)
    for rows in (1, 0):
        assert np.isfinite(outs[rows][0]).all()
        assert rel(outs[rows][0], want) <= tol
        assert rel(outs[rows][1], 3.0 * want - 4.0 * r0.astype(np.float64)) <= tol
    assert rel(outs[1][0], outs[0][0]) <= tol
    # run-to-run deterministic (fixed-order sums)
    res = torch.empty(m, dtype=dtype, device=dev)
    lo.mul(res, op, T(v, dev), 1.0, 0.0)
    assert np.array_equal(res.cpu().numpy(), outs[1][0])
