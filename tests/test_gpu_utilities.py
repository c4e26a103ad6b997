"""-m gpu: the reference's host-side diagnostics (src/utilities.jl:1-149) as CALLERS of the device `mul!` — `normest`,
`check_ctranspose`, `check_hermitian`, `check_positive_definite` — mirroring test/test_normest.jl and every use of the
check_* functions in test/test_linop.jl (:346-357 "posdef", :371-372 "Hermitian", :432-434 "Integer" on its Float64
values, :534-537) and test/test_lbfgs.jl:48-52,128-132 / test/test_lsr1.jl:30. Every product runs in libmxlo.so; the
matrices are the reference's `simple_matrix` (test/test_aux.jl:3-17: orthogonal factors around singular values
1 .. 2), rebuilt here with numpy."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

NP = {torch.float64: np.float64, torch.complex128: np.complex128}


def simple_matrix(rng, dtype, nrow, ncol, symmetric=False):
    """test/test_aux.jl:3-17"""
    npd = NP[dtype]
    if nrow == ncol == 1:
        return np.ones((1, 1), dtype=npd)

    def rand(k):
        a = rng.uniform(0, 1, (k, k))
        return (a + 1j * rng.uniform(0, 1, (k, k))).astype(npd) if dtype.is_complex else a.astype(npd)
    U, _ = np.linalg.qr(rand(nrow))
    V = U if symmetric else np.linalg.qr(rand(ncol))[0]
    S = np.zeros((nrow, ncol), dtype=npd)
    for i in range(min(nrow, ncol)):
        S[i, i] = 1 + i / (nrow - 1)
    return U @ S @ V.conj().T


def simple_vector(dtype, n):
    """test/test_aux.jl:34: T[-(-one(T))^i for i = 1:n]"""
    return np.array([-((-1.0) ** i) for i in range(1, n + 1)], dtype=NP[dtype])


def colmajor(A, dev):
    return torch.from_numpy(np.asfortranarray(A).T.copy()).to(dev).T      # a column-major torch matrix on the device


@pytest.mark.parametrize("nrow,ncol", [(10, 10), (5, 5), (3, 3), (3, 5), (5, 3), (5, 10), (10, 5)])
def test_normest(lo, dev, nrow, ncol):
    """test/test_normest.jl:8-39: |normest(S) − opnorm(Matrix(S), 2)| / opnorm <= 1e-3 for ComplexF64 and Float64 simple
    matrices (tolerance eps and the default −1), as a matrix and as LinearOperator(A); the zero matrix gives 0."""
    rng = np.random.default_rng(nrow * 100 + ncol)
    gen = torch.Generator(device=dev).manual_seed(nrow * 10 + ncol)
    for dtype in (torch.complex128, torch.float64):
        A = simple_matrix(rng, dtype, nrow, ncol)
        val = np.linalg.norm(A, 2)
        Ad = colmajor(A, dev)
        for S, tol in ((Ad, np.finfo(np.float64).eps), (lo.LinearOperatorFromMatrix(Ad), -1)):
            est, cnt = lo.normest(S, tol, 10000, generator=gen)
            assert abs(est - val) / abs(val) <= 1e-3, (dtype, est, val, cnt)
            assert cnt >= 1
    Z = torch.zeros(nrow, ncol, dtype=torch.float64, device=dev)
    for S in (Z, lo.LinearOperatorFromMatrix(Z)):
        est, cnt = lo.normest(S, -1, 10000, generator=gen)
        assert est == 0 and cnt == 0


def test_normest_drives_the_device_path(lo, dev):
    """two applies per iteration, both counted by the operator (nprod / nctprod as src/abstract.jl keeps them)"""
    rng = np.random.default_rng(3)
    op = lo.LinearOperatorFromMatrix(colmajor(simple_matrix(rng, torch.float64, 40, 30), dev))
    est, cnt = lo.normest(op, 1e-10, 500)
    assert abs(est - (1 + 29 / 39)) <= 1e-6                             # singular values 1 + i/(nrow - 1), i < 30
    assert lo.nprod(op) == cnt and lo.nctprod(op) + lo.ntprod(op) == cnt + 1


@pytest.mark.parametrize("nrow", [10, 33])
def test_posdef_testset(lo, dev, nrow):
    """test/test_linop.jl:346-357: H·diag(1:n)·H' is positive definite, H·diag(0:n-1)·H' positive semi-definite (also
    as the dense matrix of the operator)."""
    v = simple_vector(torch.complex128, nrow)
    H = lo.opHouseholder(torch.from_numpy(v / np.linalg.norm(v)).to(dev))
    lam = torch.arange(1, nrow + 1, dtype=torch.float64, device=dev).to(torch.complex128)
    op = H * lo.opDiagonal(lam) * H.H
    assert lo.check_positive_definite(op)
    assert lo.check_positive_definite(op, semi=True)
    lam0 = torch.arange(0, nrow, dtype=torch.float64, device=dev).to(torch.complex128)
    op = H * lo.opDiagonal(lam0) * H.H
    assert lo.check_positive_definite(op, semi=True)
    assert lo.check_positive_definite(lo.Matrix(op), semi=True)


def test_hermitian_testset_negatives(lo, dev):
    """test/test_linop.jl:371-372: A − A' is not Hermitian, −A'A is not positive definite; opHermitian(d, A) is both
    checked positively."""
    rng = np.random.default_rng(7)
    nrow = 10
    A = simple_matrix(rng, torch.complex128, nrow, nrow)
    d = np.real(np.diag(A)).copy()
    L = np.tril(A, -1)
    assert not lo.check_hermitian(lo.LinearOperatorFromMatrix(colmajor(L - L.conj().T, dev)))
    assert not lo.check_positive_definite(lo.LinearOperatorFromMatrix(colmajor(-(L.conj().T @ L), dev)))
    Hm = lo.opHermitian(torch.from_numpy(d).to(dev).to(torch.complex128), colmajor(L, dev))
    assert lo.check_hermitian(Hm) and lo.check_ctranspose(Hm)


def test_check_functions_on_a_general_matrix(lo, dev):
    """test/test_linop.jl:428-434 (the rounded simple matrix, as Float64 — Integer storage does not exist on the device)
    and :534-537 (a 5 x 3 ComplexF64 matrix, as matrix and as operator)."""
    rng = np.random.default_rng(11)
    A = np.round(simple_matrix(rng, torch.float64, 10, 10) * 4)
    op = lo.LinearOperatorFromMatrix(colmajor(A, dev))
    assert lo.check_ctranspose(op)
    assert lo.check_hermitian(op + op.H)
    assert lo.check_positive_definite(op * op.H, semi=True)
    B = colmajor(simple_matrix(rng, torch.complex128, 5, 3), dev)
    assert lo.check_ctranspose(B)
    assert lo.check_ctranspose(lo.LinearOperatorFromMatrix(B))
    with pytest.raises(lo.LinearOperatorException):
        lo.check_hermitian(lo.LinearOperatorFromMatrix(B))
    with pytest.raises(TypeError):
        lo.check_hermitian(lo.opEye(torch.int64, 4))


@pytest.mark.parametrize("scaling", [True, False])
def test_quasi_newton_operators_are_hermitian_positive_definite(lo, dev, scaling):
    """test/test_lbfgs.jl:33-52 (the deterministic pairs s = i·1, y = [i, 1, …, 1]) and :128-132, test/test_lsr1.jl:30."""
    n, mem = 10, 5
    B = lo.LBFGSOperator(torch.float64, n, mem=mem, scaling=scaling, device=dev)
    H = lo.InverseLBFGSOperator(torch.float64, n, mem=mem, scaling=scaling, device=dev)
    R = lo.LSR1Operator(torch.float64, n, mem=mem, scaling=scaling, device=dev)
    for op in (B, H, R):
        assert lo.check_hermitian(op)
    assert lo.check_positive_definite(B) and lo.check_positive_definite(H)
    for i in range(1, mem + 3):
        s = torch.full((n,), float(i), dtype=torch.float64, device=dev)
        y = torch.ones(n, dtype=torch.float64, device=dev)
        y[0] = float(i)
        lo.push(B, s, y)
        lo.push(H, s, y)
        lo.push(R, s, y)
    assert lo.check_positive_definite(B) and lo.check_positive_definite(H)
    for op in (B, H, R):
        assert lo.check_hermitian(op)
