"""-m gpu: sparse `LinearOperator(M::SparseMatrixCSC)` (src/constructors.jl:19-29 -> SparseArrays `mul!`) through
`mxlo_csc_*`, and sparse blocks of a fused `BlockDiagonalOperator` (test/test_linop.jl:743-756), sparse `kron` factors
(test/test_kron.jl:3-36). Parity: the C restatement of the SparseArrays loops (oracle.csc_mul) on the same inputs;
tolerance 1e-13 * (|A| |v| scale) for Float64 and 2e-6 for Float32 (rounding ORDER only: a row is summed in f64 by a lane
group with one fixed tree where the reference sums sequentially in T)."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

import oracle

pytestmark = pytest.mark.gpu

NP = {torch.float64: np.float64, torch.float32: np.float32}
TOL = {torch.float64: 1e-13, torch.float32: 2e-6}


def dev_csc(A, dev, dtype):
    """scipy CSC -> device torch.sparse_csc of the same three arrays"""
    A = sp.csc_matrix(A)
    return torch.sparse_csc_tensor(torch.from_numpy(A.indptr.astype(np.int64)), torch.from_numpy(A.indices.astype(np.int64)),
                                   torch.from_numpy(A.data.astype(NP[dtype])), size=A.shape).to(dev)


def scale_of(A, v):
    return float((abs(A) @ np.abs(v.astype(np.float64))).max()) if A.nnz else 1.0


def rand_sparse(rng, m, n, density, npd):
    A = sp.random(m, n, density, format="csc", random_state=int(rng.integers(1 << 30)), data_rvs=lambda k: rng.uniform(-1, 1, k))
    A.sort_indices()
    return A.astype(npd)


SHAPES = [(1, 1, 1.0), (7, 5, 0.5), (5, 7, 0.3), (300, 300, 0.01), (1000, 37, 0.2), (37, 1000, 0.2), (4096, 4096, 0.002),
          (2, 3000, 0.9), (3000, 2, 0.9), (513, 129, 0.0), (10_000, 10_000, 3e-4)]


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("m,n,density", SHAPES)
def test_csc_mul_vs_oracle(lo, dev, dtype, m, n, density):
    """prod! / tprod! / ctprod! for every α, β kind: β = 0 on NaN-filled res (must not propagate), β = 1, general;
    Float32 data with Float64 scalars (MXLO_ALPHA_F64 / _BETA_F64, like every other leaf)."""
    npd = NP[dtype]
    rng = np.random.default_rng(m * 31 + n)
    A = rand_sparse(rng, m, n, density, npd)
    op = lo.LinearOperatorFromMatrix(dev_csc(A, dev, dtype))
    assert op.shape == (m, n)
    for trans in (False, True):
        nin, nout = (m, n) if trans else (n, m)
        v = rng.uniform(-1, 1, nin).astype(npd)
        r0 = rng.uniform(-1, 1, nout).astype(npd)
        for a, b in ((1.0, 0.0), (2.0, -3.0), (0.0, 1.0), (-0.5, 1.0), (np.float32(1.5), np.float32(0.25))):
            start = np.full(nout, np.nan, npd) if b == 0 else r0
            res = torch.from_numpy(start.copy()).to(dev)
            lo.mul(res, lo.transpose(op) if trans else op, torch.from_numpy(v).to(dev), a, b)
            flags = 0
            if dtype == torch.float32 and not isinstance(a, np.float32):
                flags = 0x1 | 0x8
            want = oracle.csc_mul(np.zeros(nout, npd) if b == 0 else r0.copy(), A.indptr + 1, A.indices + 1, A.data, m, n, v,
                                  float(a), float(b), trans=trans, flags=flags)
            tol = TOL[dtype] * (abs(float(a)) * scale_of(A.T if trans else A, v) + abs(float(b)) + 1e-300)
            got = res.cpu().numpy()
            assert np.isfinite(got).all()
            assert np.abs(got.astype(np.float64) - want.astype(np.float64)).max() <= tol, (trans, a, b)
    # adjoint of a real sparse operator == transpose (src/adjtrans.jl)
    u = rng.uniform(-1, 1, m).astype(npd)
    ra, rt = (torch.empty(n, dtype=dtype, device=dev) for _ in range(2))
    lo.mul(ra, lo.adjoint(op), torch.from_numpy(u).to(dev), 1.0, 0.0)
    lo.mul(rt, lo.transpose(op), torch.from_numpy(u).to(dev), 1.0, 0.0)
    assert torch.equal(ra, rt)


def test_csc_row_lengths_from_one_lane_to_a_full_wave(lo, dev):
    """The lane group per row is chosen from the mean row length: exercise every group width (1 … 64 lanes), rows far
    longer than the group (several trips) and empty rows next to full ones; each apply twice (run-to-run bit identity)."""
    rng = np.random.default_rng(3)
    for mean in (1, 2, 3, 6, 12, 24, 48, 100, 700):
        m, n = 2000, 1500
        rows = []
        for i in range(m):
            k = 0 if i % 7 == 3 else min(n, int(rng.integers(max(1, mean // 2), 2 * mean + 1)))
            rows.append(np.sort(rng.choice(n, size=k, replace=False)))
        indptr = np.concatenate([[0], np.cumsum([r.size for r in rows])])
        data = rng.uniform(-1, 1, indptr[-1])
        A = sp.csr_matrix((data, np.concatenate(rows) if indptr[-1] else np.zeros(0, int), indptr), shape=(m, n)).tocsc()
        A.sort_indices()
        op = lo.LinearOperatorFromMatrix(dev_csc(A, dev, torch.float64))
        info = op._csc.info()
        assert info["nnz"] == A.nnz and info["chunk"] == 2048 and info["chunks_n"] >= A.nnz // 2048
        v = rng.uniform(-1, 1, n)
        res, res2 = torch.empty(m, dtype=torch.float64, device=dev), torch.empty(m, dtype=torch.float64, device=dev)
        lo.mul(res, op, torch.from_numpy(v).to(dev), 1.0, 0.0)
        lo.mul(res2, op, torch.from_numpy(v).to(dev), 1.0, 0.0)
        assert torch.equal(res, res2)
        want = oracle.csc_mul(np.zeros(m), A.indptr + 1, A.indices + 1, A.data, m, n, v, 1.0, 0.0)
        assert np.abs(res.cpu().numpy() - want).max() <= 1e-13 * scale_of(A, v), mean
        u = rng.uniform(-1, 1, m)
        rt = torch.empty(n, dtype=torch.float64, device=dev)
        lo.mul(rt, lo.transpose(op), torch.from_numpy(u).to(dev), 1.0, 0.0)
        want = oracle.csc_mul(np.zeros(n), A.indptr + 1, A.indices + 1, A.data, m, n, u, 1.0, 0.0, trans=True)
        assert np.abs(rt.cpu().numpy() - want).max() <= 1e-13 * scale_of(A.T, u), mean


def test_csr_and_coo_layouts_and_unsorted_duplicate_entries(lo, dev):
    """A `torch.sparse_csr` tensor is the CSC storage of the transpose: aliased with N and T swapped (no conversion).
    COO is converted once. Unsorted rows within a column and duplicate (row, col) entries are legal for the
    reference's loops (each stored entry contributes): same here."""
    rng = np.random.default_rng(8)
    A = rand_sparse(rng, 60, 45, 0.2, np.float64)
    v, u = rng.uniform(-1, 1, 45), rng.uniform(-1, 1, 60)
    csc = dev_csc(A, dev, torch.float64)
    for M in (csc.to_sparse_csr(), csc.to_sparse_coo(), A, sp.coo_matrix(A)):          # … and scipy.sparse matrices from the host
        op = lo.LinearOperatorFromMatrix(M)
        got = (op * torch.from_numpy(v).to(dev)).cpu().numpy()
        assert np.abs(got - A @ v).max() <= 1e-13 * scale_of(A, v)
        got = (lo.transpose(op) * torch.from_numpy(u).to(dev)).cpu().numpy()
        assert np.abs(got - A.T @ u).max() <= 1e-13 * scale_of(A.T, u)
    # hand-built: column 0 holds rows (2, 0, 2) — unsorted with a duplicate; column 1 empty; column 2 rows (1,)
    colptr = np.array([1, 4, 4, 5]); rowval = np.array([3, 1, 3, 2]); nz = np.array([1.5, -2.0, 0.25, 4.0])
    M = lo.sparse_csc(colptr, rowval, nz, 3, 3, index_base=1, device=dev)
    op = lo.LinearOperatorFromMatrix(M)
    x = np.array([2.0, 5.0, -1.0])
    want = oracle.csc_mul(np.zeros(3), colptr, rowval, nz, 3, 3, x, 1.0, 0.0)
    assert np.array_equal((op * torch.from_numpy(x).to(dev)).cpu().numpy(), want)          # tiny sums: exact
    want = oracle.csc_mul(np.zeros(3), colptr, rowval, nz, 3, 3, x, 1.0, 0.0, trans=True)
    assert np.array_equal((lo.transpose(op) * torch.from_numpy(x).to(dev)).cpu().numpy(), want)


def test_values_are_aliased_and_updates_are_seen(lo, dev):
    """The reference's closure holds M itself: `nonzeros(M) .*= 2` changes the operator. Here Aᵀ*x reads the values in
    place and A*x re-gathers its row-ordered snapshot when the values tensor's version moved (in-place torch ops bump
    it; `lo.touched` for writes torch cannot see)."""
    rng = np.random.default_rng(12)
    A = rand_sparse(rng, 300, 200, 0.05, np.float64)
    M = dev_csc(A, dev, torch.float64)
    op = lo.LinearOperatorFromMatrix(M)
    v = torch.from_numpy(rng.uniform(-1, 1, 200)).to(dev)
    u = torch.from_numpy(rng.uniform(-1, 1, 300)).to(dev)
    y0, z0 = (op * v).clone(), (lo.transpose(op) * u).clone()
    M.values().mul_(2.0)
    assert torch.allclose(op * v, 2 * y0, rtol=1e-15, atol=0) and torch.allclose(lo.transpose(op) * u, 2 * z0, rtol=1e-15, atol=0)
    raw = M.values()
    raw.view(torch.int64).copy_((raw * 0.25).view(torch.int64))     # a write through a reinterpreting view still bumps
    lo.touched(raw)
    assert torch.allclose(op * v, 0.5 * y0, rtol=1e-15, atol=0)


def test_blockdiagonal_with_operator_matrix_and_sparse_blocks(lo, dev):
    """test/test_linop.jl:739-756: `BlockDiagonalOperator(A, B, C)` with A an operator built from a closure (there: a
    Cholesky solve = diag(0.5, 0.25, 0.125)), B = rand(4, 2), C = sprand(2, 4, 0.5), compared with the dense block
    matrix for M, transpose(M), M' to sqrt(eps)·‖D‖. The closure block makes the operator take the reference's per-block
    loop; the second operator (diagonal leaf instead of the closure) is ONE fused launch with a sparse block."""
    rng = np.random.default_rng(5)
    dinv = torch.tensor([0.5, 0.25, 0.125], dtype=torch.float64, device=dev)
    closure = lo.LinearOperator(torch.float64, 3, 3, True, True, lambda y, v: y.copy_(dinv * v), S=lo.Storage(torch.float64, dev))
    B = rng.uniform(0, 1, (4, 2))
    Cs = sp.random(2, 4, 0.5, format="csc", random_state=3)
    D = np.zeros((9, 9))
    D[:3, :3] = np.diag([0.5, 0.25, 0.125]); D[3:7, 3:5] = B; D[7:9, 5:9] = Cs.toarray()
    Bd = torch.from_numpy(np.ascontiguousarray(B.T)).to(dev).t()        # column-major, Julia's Matrix layout
    Cd = dev_csc(Cs, dev, torch.float64)
    for first, fused in ((closure, False), (lo.opDiagonal(dinv), True)):
        M = lo.BlockDiagonalOperator(first, Bd, Cd)
        assert M.shape == (9, 9)
        assert hasattr(M, "_keepalive") == fused
        tol = np.sqrt(np.finfo(np.float64).eps) * np.linalg.norm(D)
        assert np.linalg.norm(lo.Matrix(M).cpu().numpy() - D) <= tol
        assert np.linalg.norm(lo.Matrix(lo.transpose(M)).cpu().numpy() - D.T) <= tol
        assert np.linalg.norm(lo.Matrix(lo.adjoint(M)).cpu().numpy() - D.T) <= tol


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_fused_blockdiagonal_mix_with_sparse_blocks_vs_per_block_applies(lo, dev, dtype):
    """Many blocks of every fusable kind incl. sparse ones at odd offsets: the one-launch operator against the sum of its
    blocks applied one by one (the reference's loop, src/special-operators.jl:258-289), N / T, β = 0 and β ≠ 0; the
    sparse blocks' results are bit-identical to the standalone sparse leaf (same kernel body, same lane groups)."""
    npd = NP[dtype]
    rng = np.random.default_rng(77)
    blocks, dense = [], []
    for k in range(40):
        kind = k % 4
        if kind == 0:
            n = int(rng.integers(1, 700))
            d = rng.uniform(0.5, 1.5, n).astype(npd)
            blocks.append(lo.opDiagonal(torch.from_numpy(d).to(dev))); dense.append(sp.diags(d.astype(np.float64)))
        elif kind == 1:
            m, n = int(rng.integers(1, 60)), int(rng.integers(1, 60))
            B = rng.uniform(-1, 1, (m, n)).astype(npd)
            blocks.append(torch.from_numpy(np.asfortranarray(B).T.copy()).to(dev).t()); dense.append(sp.csc_matrix(B.astype(np.float64)))
        elif kind == 2:
            m, n = int(rng.integers(1, 900)), int(rng.integers(1, 900))
            A = rand_sparse(rng, m, n, float(rng.uniform(0.0, 0.08)), npd)
            blocks.append(dev_csc(A, dev, dtype)); dense.append(A.astype(np.float64))
        else:
            n = int(rng.integers(1, 300))
            blocks.append(lo.opEye(dtype, n, S=lo.Storage(dtype, dev))); dense.append(sp.identity(n, format="csc"))
    M = lo.BlockDiagonalOperator(*blocks)
    assert hasattr(M, "_keepalive"), "every block kind here is fusable: one launch expected"
    D = sp.block_diag(dense, format="csc")
    assert M.shape == D.shape
    tol = 1e-12 if dtype == torch.float64 else 2e-5
    for trans in (False, True):
        op, Dm = (lo.transpose(M), D.T) if trans else (M, D)
        nout, nin = Dm.shape
        v = rng.uniform(-1, 1, nin).astype(npd)
        r0 = rng.uniform(-1, 1, nout).astype(npd)
        for a, b in ((1.0, 0.0), (2.0, -3.0)):
            res = torch.from_numpy((np.full(nout, np.nan, npd) if b == 0 else r0).copy()).to(dev)
            lo.mul(res, op, torch.from_numpy(v).to(dev), a, b)
            want = a * (Dm @ v.astype(np.float64)) + (b * r0.astype(np.float64) if b != 0 else 0)
            assert np.abs(res.cpu().numpy() - want).max() <= tol * (abs(a) * scale_of(sp.csc_matrix(Dm), v) + abs(b))
    # a sparse block inside the fused operator == the standalone leaf, bit for bit
    k = 2
    off_r = sum(d.shape[0] for d in dense[:k]); off_c = sum(d.shape[1] for d in dense[:k])
    m, n = dense[k].shape
    x = torch.zeros(D.shape[1], dtype=dtype, device=dev)
    xs = torch.from_numpy(rng.uniform(-1, 1, n).astype(npd)).to(dev)
    x[off_c:off_c + n] = xs
    full = M * x
    alone = lo.LinearOperatorFromMatrix(blocks[k]) * xs
    assert torch.equal(full[off_r:off_r + m], alone)


def test_kron_with_sparse_factors(lo, dev):
    """test/test_kron.jl:3-36 runs kron over dense AND sparse factors (simple_sparse_matrix(Float64, 10, 10)):
    `kron(LinearOperator(A), B)` with A sparse — the factor is materialised with Matrix(op) like any operator factor —
    against numpy's kron within the reference's eps·‖K‖₁-class criterion, for K, K', transpose(K)."""
    rng = np.random.default_rng(10)
    A = rand_sparse(rng, 10, 10, 0.3, np.float64)
    for B in (rng.uniform(-1, 1, (2, 3)), rand_sparse(rng, 10, 10, 0.3, np.float64)):
        Bn = B.toarray() if sp.issparse(B) else B
        Bd = lo.LinearOperatorFromMatrix(dev_csc(B, dev, torch.float64)) if sp.issparse(B) else torch.from_numpy(Bn).to(dev)
        K = np.kron(A.toarray(), Bn)
        T1 = lo.kron(lo.LinearOperatorFromMatrix(dev_csc(A, dev, torch.float64)), Bd)
        normK = np.abs(K).sum(axis=0).max()
        assert np.abs(lo.Matrix(T1).cpu().numpy() - K).sum(axis=0).max() <= 1e-12 * normK
        assert np.abs(lo.Matrix(lo.transpose(T1)).cpu().numpy() - K.T).sum(axis=0).max() <= 1e-12 * normK
        assert np.abs(lo.Matrix(lo.adjoint(T1)).cpu().numpy() - K.T).sum(axis=0).max() <= 1e-12 * normK


def test_csc_create_validates_like_the_sparsematrixcsc_constructor(lo, dev):
    """SparseMatrixCSC's inner constructor rejects a colptr that does not start at 1 or decreases and row indices
    outside 1:m (ArgumentError); mxlo_csc_create returns MXLO_EINVAL naming the offending position."""
    i64 = lambda a: torch.tensor(a, dtype=torch.int64, device=dev)
    vals = torch.ones(3, dtype=torch.float64, device=dev)
    bad = [((3, 2), [1, 1, 3], [0, 1, 2], "colptr"),            # does not start at 0 (index_base 0)
           ((3, 2), [0, 2, 1], [0, 1, 2], "decreases"),
           ((3, 2), [0, 2, 3], [0, 3, 2], "row index")]
    for shape, cp, rv, word in bad:
        M = torch.sparse_csc_tensor(i64(cp), i64(rv), vals, size=shape, check_invariants=False)
        with pytest.raises(Exception, match=word):
            lo.LinearOperatorFromMatrix(M)
    # empty matrix and empty pattern
    for shape in ((0, 0), (0, 4), (5, 0), (3, 3)):
        cp = i64([0] * (shape[1] + 1))
        M = torch.sparse_csc_tensor(cp, i64([]), torch.zeros(0, dtype=torch.float64, device=dev), size=shape)
        op = lo.LinearOperatorFromMatrix(M)
        res = torch.full((shape[0],), float("nan"), dtype=torch.float64, device=dev)
        lo.mul(res, op, torch.ones(shape[1], dtype=torch.float64, device=dev), 1.0, 0.0)
        assert bool((res == 0).all())


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_rows_and_columns_longer_than_a_chunk(lo, dev, dtype):
    """A few very long rows / columns among many short ones (an arrow-head pattern): rows of 513 … 2048 entries take a
    chunk of their own, longer ones are summed piecewise with the second (fix-up) launch. Such a matrix cannot be a
    block of the ONE-launch block-diagonal operator: the host mirror then takes the per-block loop — same numbers."""
    npd = NP[dtype]
    rng = np.random.default_rng(21)
    n = 30_000
    nb = 180_000                                           # background: ~6 entries per row, duplicates summed
    rows, cols, vals = [rng.integers(0, n, nb)], [rng.integers(0, n, nb)], [rng.uniform(-1, 1, nb)]
    for r, cnt in ((0, n), (17, 600), (999, 2030), (1000, 2070), (n - 1, 20_001)):       # 2048 = one chunk
        cs = rng.choice(n, size=cnt, replace=False)
        rows.append(np.full(cnt, r)); cols.append(cs); vals.append(rng.uniform(-1, 1, cnt))
    for c, cnt in ((5, n), (n // 2, 9000)):
        rs = rng.choice(n, size=cnt, replace=False)
        rows.append(rs); cols.append(np.full(cnt, c)); vals.append(rng.uniform(-1, 1, cnt))
    A = sp.csc_matrix(sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))).astype(npd)
    A.sum_duplicates()
    A.sort_indices()
    op = lo.LinearOperatorFromMatrix(dev_csc(A, dev, dtype))
    inf = op._csc.info()
    assert inf["long_rows"] == 3 and inf["long_cols"] == 2          # rows of ~2075, ~20005 and n entries; two long columns
    assert getattr(op, "_leaf", None) is None
    for trans in (False, True):
        v = rng.uniform(-1, 1, n).astype(npd)
        r0 = rng.uniform(-1, 1, n).astype(npd)
        for a, b in ((1.0, 0.0), (2.0, -3.0)):
            res = torch.from_numpy((np.full(n, np.nan, npd) if b == 0 else r0).copy()).to(dev)
            res2 = res.clone()
            o = lo.transpose(op) if trans else op
            lo.mul(res, o, torch.from_numpy(v).to(dev), a, b)
            lo.mul(res2, o, torch.from_numpy(v).to(dev), a, b)
            assert torch.equal(res, res2)
            want = a * ((A.T if trans else A).astype(np.float64) @ v.astype(np.float64)) + (b * r0.astype(np.float64) if b else 0)
            tol = (1e-12 if dtype == torch.float64 else 2e-5) * (abs(a) * scale_of(A.T if trans else A, v) + abs(b))
            assert np.abs(res.cpu().numpy() - want).max() <= tol, (trans, a, b)
    d = torch.from_numpy(rng.uniform(0.5, 1.5, 100).astype(npd)).to(dev)
    M = lo.BlockDiagonalOperator(lo.opDiagonal(d), op)
    assert not hasattr(M, "_keepalive")
    x = rng.uniform(-1, 1, n + 100).astype(npd)
    got = (M * torch.from_numpy(x).to(dev)).cpu().numpy()
    want = np.concatenate([d.cpu().numpy().astype(np.float64) * x[:100], A.astype(np.float64) @ x[100:].astype(np.float64)])
    assert np.abs(got - want).max() <= (1e-12 if dtype == torch.float64 else 2e-5) * scale_of(A, x[100:])


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_ophermitian_of_a_sparse_matrix(lo, dev, dtype):
    """`opHermitian(d, A)` / `opHermitian(A)` (src/linalg.jl:105-127) accept any AbstractMatrix; for a sparse A the
    strict lower triangle `tril(A, -1)` is sparse. Against the dense symmetric matrix d + L + Lᵀ (how
    test/test_linop.jl:360-380 checks the dense form), whatever sits on or above A's diagonal must be ignored, α / β."""
    npd = NP[dtype]
    rng = np.random.default_rng(41)
    n = 700
    A = rand_sparse(rng, n, n, 0.02, npd)
    A.setdiag(rng.uniform(1, 2, n).astype(npd))
    A = sp.csc_matrix(A)
    L = sp.tril(A, -1).toarray().astype(np.float64)
    d = rng.uniform(-1, 1, n).astype(npd)
    for op, diag in ((lo.opHermitian(torch.from_numpy(d).to(dev), dev_csc(A, dev, dtype)), d.astype(np.float64)),
                     (lo.opHermitian(dev_csc(A, dev, dtype)), A.diagonal().astype(np.float64))):
        assert lo.issymmetric(op) and lo.ishermitian(op) and op.shape == (n, n)
        H = np.diag(diag) + L + L.T
        v = rng.uniform(-1, 1, n).astype(npd)
        r0 = rng.uniform(-1, 1, n).astype(npd)
        for a, b in ((1.0, 0.0), (2.0, -3.0)):
            res = torch.from_numpy((np.full(n, np.nan, npd) if b == 0 else r0).copy()).to(dev)
            lo.mul(res, op, torch.from_numpy(v).to(dev), a, b)
            want = a * (H @ v.astype(np.float64)) + (b * r0.astype(np.float64) if b else 0)
            tol = (1e-12 if dtype == torch.float64 else 2e-5) * (abs(a) * float((np.abs(H) @ np.abs(v)).max()) + abs(b))
            assert np.abs(res.cpu().numpy() - want).max() <= tol


@pytest.mark.parametrize("dtype,tol", [(torch.complex128, 1e-13), (torch.complex64, 5e-6)])
def test_complex_sparse_operator_native_and_through_real_planes(lo, dev, dtype, tol, monkeypatch):
    """test/test_linop.jl:44 and test/test_cat.jl:5-25 build operators from `simple_sparse_matrix(ComplexF64, …)`. The
    native instantiation (`mxlo_csc_mul_c`: the chunked sweep on complex elements, mode C conjugating the stored values)
    against the oracle's restatement of the SparseArrays loops, against the dense matrix, and against the real-planes form
    (four real sweeps between a split and a join pass — the independent device implementation, MXLO_SPARSE_COMPLEX_PLANES=1):
    M*v, transpose(M)*u, M'*u, Real and Complex α, β, β = 0 on NaN, CSC and CSR (transposed alias) storage, a row beyond
    a chunk."""
    rng = np.random.default_rng(63)
    cdt = np.complex128 if dtype == torch.complex128 else np.complex64
    for m, n, dens, long_row in ((30, 20, 0.3, False), (400, 700, 0.02, False), (50, 3000, 0.01, True)):
        A = sp.random(m, n, dens, format="lil", random_state=7).astype(np.complex128)
        if long_row:
            A[3, :] = 1.0                                          # 3000 entries in one row: pieces + fix-up launch
        A = sp.csc_matrix(A)
        A.data = (rng.standard_normal(A.nnz) + 1j * rng.standard_normal(A.nnz))
        A = A.astype(cdt)
        A.sort_indices()
        M = torch.sparse_csc_tensor(torch.from_numpy(A.indptr.astype(np.int64)), torch.from_numpy(A.indices.astype(np.int64)),
                                    torch.from_numpy(A.data), size=A.shape).to(dev)
        ops = {"native csc": lo.LinearOperatorFromMatrix(M), "native csr alias": lo.LinearOperatorFromMatrix(M.to_sparse_csr())}
        monkeypatch.setenv("MXLO_SPARSE_COMPLEX_PLANES", "1")
        ops["planes"] = lo.LinearOperatorFromMatrix(M)
        monkeypatch.delenv("MXLO_SPARSE_COMPLEX_PLANES")
        assert hasattr(ops["native csc"], "_csc") and not hasattr(ops["planes"], "_csc")
        D = A.toarray().astype(np.complex128)
        for mode, Dm in ((False, D), ("T", D.T), ("C", D.conj().T)):
            nout, nin = Dm.shape
            v = (rng.standard_normal(nin) + 1j * rng.standard_normal(nin)).astype(cdt)
            r0 = (rng.standard_normal(nout) + 1j * rng.standard_normal(nout)).astype(cdt)
            for a, b in ((1.0, 0.0), (1.5 - 0.5j, 0.25 + 2j), (2.0, -3.0), (-1j, 1.0)):
                a_real, b_real = not isinstance(a, complex), not isinstance(b, complex)
                flags = (0x20 if a_real else 0) | (0x40 if b_real else 0) | (0x1 | 0x8)
                want_o = oracle.csc_mul(np.zeros(nout, cdt) if b == 0 else r0.copy(), A.indptr + 1, A.indices + 1, A.data, m, n, v,
                                        a, b, trans=mode, flags=flags)
                want_d = a * (Dm @ v.astype(np.complex128)) + (b * r0.astype(np.complex128) if b != 0 else 0)
                scale = abs(a) * float((np.abs(Dm) @ np.abs(v)).max()) + abs(b) * float(np.abs(r0).max())
                got = {}
                for name, op in ops.items():
                    o = op if mode is False else (lo.transpose(op) if mode == "T" else lo.adjoint(op))
                    res = torch.from_numpy((np.full(nout, np.nan + 0j, cdt) if b == 0 else r0).copy()).to(dev)
                    lo.mul(res, o, torch.from_numpy(v).to(dev), a, b)
                    got[name] = res.cpu().numpy()
                    assert np.abs(got[name] - want_d).max() <= 8 * tol * scale, (name, mode, a, b)
                    assert np.abs(got[name].astype(np.complex128) - want_o.astype(np.complex128)).max() <= 8 * tol * scale, (name, mode)
        # in-place value updates are seen (A*x through the refreshed snapshot, A'*x at once)
        op = ops["native csc"]
        v = torch.from_numpy((rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(cdt)).to(dev)
        y0 = (op * v).clone()
        M.values().mul_(2.0)
        assert torch.allclose(op * v, 2 * y0, rtol=1e-6 if dtype == torch.complex64 else 1e-14, atol=0)


def test_warmed_sparse_applies_only_launch_kernels(lo, dev):
    """The allocation / synchronisation contract (tests/test_gpu_contract.py; test/test_linop_allocs.jl in the reference):
    a warmed sparse `mul!` — the leaf in both modes, a row longer than a chunk (two launches), and the fused
    block-diagonal operator with sparse blocks — issues kernel launches and nothing else; ONE launch for the leaf."""
    import ctypes as C
    names = ("malloc", "free", "h2d", "d2h", "d2d", "d2h_bytes", "stream_sync", "device_sync", "event_sync", "memset_async",
             "launch", "blocking_copy")

    def snap():
        a = (C.c_int64 * 12)()
        lo._lib.call("mxlo_debug_counters", a)
        return dict(zip(names, list(a)))

    rng = np.random.default_rng(2)
    A = rand_sparse(rng, 5000, 4000, 0.004, np.float64)
    op = lo.LinearOperatorFromMatrix(dev_csc(A, dev, torch.float64))
    B = sp.lil_matrix((3000, 3000)); B[7, :] = 1.0; B.setdiag(2.0)
    long_op = lo.LinearOperatorFromMatrix(dev_csc(sp.csc_matrix(B), dev, torch.float64))
    bd = lo.BlockDiagonalOperator(lo.opDiagonal(torch.rand(100, dtype=torch.float64, device=dev)), dev_csc(A, dev, torch.float64))
    assert hasattr(bd, "_keepalive")
    cases = [("A*x", op, 4000, 5000, 1), ("A'*x", lo.transpose(op), 5000, 4000, 1), ("row > chunk", long_op, 3000, 3000, 2),
             ("fused block-diagonal", bd, 4100, 5100, 1)]
    for name, o, nin, nout, launches in cases:
        v = torch.rand(nin, dtype=torch.float64, device=dev)
        res = torch.rand(nout, dtype=torch.float64, device=dev)
        for _ in range(3):
            lo.mul(res, o, v, 2.0, -3.0)
        import gc
        gc.collect()
        torch.cuda.synchronize()
        a = snap()
        for _ in range(4):
            lo.mul(res, o, v, 2.0, -3.0)
        b = snap()
        d = {k: b[k] - a[k] for k in names}
        assert not {k: x for k, x in d.items() if k != "launch" and x}, (name, d)
        assert d["launch"] == 4 * launches, (name, d)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_sparse_block_apply_reads_the_matrix_once_and_matches_the_columns(lo, dev, dtype):
    """`mul!(res::Matrix, op, V::Matrix, α, β)` (src/operations.jl:34-36; test/test_linop.jl:64-76 applies operators to
    `hcat(v, -2v)`): for a sparse operator the block goes through `mxlo_csc_mul_block` — the chunks are staged in LDS once
    per group of 8 columns — and every column must equal the single-vector apply BIT FOR BIT (same walk, same order), for
    k = 1 … 11 (more than one group), both modes, β = 0 on NaN and β ≠ 0, column-major operands with a padded leading
    dimension, and a matrix with rows beyond a chunk (the piecewise path has per-column carry slots)."""
    npd = NP[dtype]
    rng = np.random.default_rng(88)
    m, n = 3000, 2600
    base = rand_sparse(rng, m, n, 0.004, npd).tolil()
    base[5, :] = rng.uniform(-1, 1, n)                     # a row of 2600 entries: two pieces
    base[:, 9] = rng.uniform(-1, 1, (m, 1))                # a column of 3000 entries
    A = sp.csc_matrix(base).astype(npd)
    A.sort_indices()
    op = lo.LinearOperatorFromMatrix(dev_csc(A, dev, dtype))
    for o, nin, nout in ((op, n, m), (lo.transpose(op), m, n)):
        for k in (1, 2, 8, 11):
            Vfull = torch.from_numpy(rng.uniform(-1, 1, (k, nin + 3)).astype(npd)).to(dev)          # leading dimension nin + 3
            R0 = torch.from_numpy(rng.uniform(-1, 1, (k, nout + 5)).astype(npd)).to(dev)
            V = Vfull[:, :nin].t()                                                                  # nin x k, column-major
            for a, b in ((1.0, 0.0), (2.0, -3.0)):
                Rblk = (torch.full_like(R0, float("nan")) if b == 0 else R0.clone())[:, :nout].t()
                Rcol = (torch.full_like(R0, float("nan")) if b == 0 else R0.clone())[:, :nout].t()
                lo.mul(Rblk, o, V, a, b)
                for j in range(k):
                    lo.mul(Rcol[:, j], o, V[:, j], a, b)
                assert torch.equal(Rblk, Rcol), (k, a, b)
            Dm = (A.T if o is not op else A).astype(np.float64)
            want = Dm @ V.cpu().numpy().astype(np.float64)
            got = (o * V) if hasattr(o, "__mul__") else None
            res = torch.empty(k, nout, dtype=dtype, device=dev).t()
            lo.mul(res, o, V, 1.0, 0.0)
            tol = (1e-12 if dtype == torch.float64 else 2e-5) * float((abs(Dm) @ np.abs(V.cpu().numpy().astype(np.float64))).max())
            assert np.abs(res.cpu().numpy() - want).max() <= tol


def test_kat_vcat_of_eye_and_a_sparse_identity(lo, dev):
    """test/test_cat.jl:46-48 and :193-195: `K = [opEye(2); sparse(1.0I, 2, 2)]; v = simple_vector(Float64, 2);
    @test all(K * v .== [v; v])` — exact arithmetic, so equality is bitwise; also the transposed and adjoint forms of the
    same identity (`[opEye(2) sparse(I)]'`), α, β = 3, −4 as test/test_cat.jl:96-118 uses them, and hcat."""
    S = lo.Storage(torch.float64, dev)
    I2 = lo.sparse_csc([1, 2, 3], [1, 2], [1.0, 1.0], 2, 2, index_base=1, device=dev)        # sparse(1.0I, 2, 2) as Julia stores it
    K = lo.vcat(lo.opEye(torch.float64, 2, S=S), I2)
    v = torch.tensor([1.0, -1.0], dtype=torch.float64, device=dev)                           # simple_vector(Float64, 2)
    assert torch.equal(K * v, torch.cat([v, v]))
    w = torch.tensor([1.0, -1.0, 1.0, -1.0], dtype=torch.float64, device=dev)
    assert torch.equal(lo.transpose(K) * w, 2 * v) and torch.equal(lo.adjoint(K) * w, 2 * v)
    res = torch.tensor([0.5, 0.25, 2.0, -8.0], dtype=torch.float64, device=dev)
    lo.mul(res, K, v, 3.0, -4.0)
    assert torch.equal(res, torch.tensor([3.0 - 2.0, -3.0 - 1.0, 3.0 - 8.0, -3.0 + 32.0], dtype=torch.float64, device=dev))
    H = lo.hcat(lo.opEye(torch.float64, 2, S=S), I2)
    assert torch.equal(H * w, torch.tensor([2.0, -2.0], dtype=torch.float64, device=dev))
