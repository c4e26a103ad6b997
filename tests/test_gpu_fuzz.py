"""-m gpu: randomized differential test of the host mirror + kernels. Random operator trees are grown from every
leaf of the hot path (opDiagonal square/rectangular, opEye, opZeros, opOnes, dense and sparse matrices, opHouseholder,
opHermitian, restriction/extension, BlockDiagonalOperator, kron, forward/inverse L-BFGS, L-SR1) with the
reference's combinators (+, -, *, scalar, transpose, adjoint, hcat, vcat, op[rows, cols]) and compared with the
same tree evaluated on dense NumPy matrices: op*v, op'*w, the 5-arg form with (α, β) = (3, -4), and Matrix(op).
Tolerance 1e-10 relative to ||M||·||v|| (fp64; trees can be badly scaled, leaves are tested tightly elsewhere)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def TM(A, dev):
    return torch.from_numpy(np.asfortranarray(A).T.copy()).to(dev).t()     # column-major device matrix


class D:
    """Dense image M of a subtree together with G >= |every intermediate| (same tree on absolute values): the
    forward-error scale of the operator arithmetic, which stays meaningful when M itself cancels to zero."""

    def __init__(self, M, G=None):
        self.M, self.G = np.asarray(M, float), np.abs(M) if G is None else G

    shape = property(lambda s: s.M.shape)
    T = property(lambda s: D(s.M.T, s.G.T))
    __add__ = lambda a, b: D(a.M + b.M, a.G + b.G)
    __sub__ = lambda a, b: D(a.M - b.M, a.G + b.G)
    __matmul__ = lambda a, b: D(a.M @ b.M, a.G @ b.G)
    scaled = lambda a, x: D(x * a.M, abs(x) * a.G)
    pick = lambda a, r, c: D(a.M[np.ix_(r, c)], a.G[np.ix_(r, c)])


def hstack(a, b):
    return D(np.hstack([a.M, b.M]), np.hstack([a.G, b.G]))


def vstack(a, b):
    return D(np.vstack([a.M, b.M]), np.vstack([a.G, b.G]))


class Gen:
    def __init__(self, lo, dev, seed):
        self.lo, self.dev, self.rng = lo, dev, np.random.default_rng(seed)

    def csc(self, A):
        """device torch.sparse_csc of the nonzero pattern of the dense array A"""
        import scipy.sparse as sp
        S = sp.csc_matrix(A)
        return torch.sparse_csc_tensor(torch.from_numpy(S.indptr.astype(np.int64)), torch.from_numpy(S.indices.astype(np.int64)),
                                       torch.from_numpy(S.data.astype(np.float64)), size=A.shape).to(self.dev)

    def leaf(self, m, n):
        """A random leaf of shape (m, n) as (operator, dense, hN, hT, description). hN / hT: does prod! / tprod! honour
        the caller's α and β on EVERY output row? The reference's restrictions ignore both (special-operators.jl:
        167-174); rectangular opEye / opDiagonal overwrite the rows past min(nrow, ncol) whatever β is (:36-44,
        :144-151). The mirror reproduces this, so the dense model has to know it."""
        lo, dev, rng = self.lo, self.dev, self.rng
        kinds = ["dense", "zeros", "ones", "diag_rect", "eye_rect", "sparse"]
        if m == n:
            kinds += ["diag", "householder", "hermitian", "lbfgs", "invlbfgs", "lsr1", "blockdiag"]
            if m % 2 == 0 and m >= 4:
                kinds.append("kron")
        if m < n:
            kinds.append("restriction")
        if m > n:
            kinds.append("extension")
        k = kinds[rng.integers(len(kinds))]
        S = lo.Storage(torch.float64, dev)
        if k == "dense":
            A = rng.standard_normal((m, n))
            return lo.LinearOperatorFromMatrix(TM(A, dev)), D(A), True, True, f"{k}[{m}x{n}]"
        if k == "sparse":                                           # LinearOperator(M::SparseMatrixCSC): mxlo_csc_*
            A = rng.standard_normal((m, n)) * (rng.random((m, n)) < rng.uniform(0.05, 0.6))
            return lo.LinearOperatorFromMatrix(self.csc(A)), D(A), True, True, f"{k}[{m}x{n}]"
        if k == "zeros":
            return lo.opZeros(torch.float64, m, n, S=S), D(np.zeros((m, n))), True, True, f"{k}[{m}x{n}]"
        if k == "ones":
            return lo.opOnes(torch.float64, m, n, S=S), D(np.ones((m, n))), True, True, f"{k}[{m}x{n}]"
        if k == "diag_rect":
            d = rng.standard_normal(min(m, n))
            return lo.opDiagonal(m, n, T(d, dev)), D(np.eye(m, n) * np.pad(d, (0, max(0, n - len(d))))[None, :n]), m <= n, n <= m, f"{k}[{m}x{n}]"
        if k == "eye_rect":
            return lo.opEye(torch.float64, m, n, S=S), D(np.eye(m, n)), m <= n, n <= m, f"{k}[{m}x{n}]"
        if k == "diag":
            d = rng.standard_normal(n)
            return lo.opDiagonal(T(d, dev)), D(np.diag(d)), True, True, f"{k}[{m}x{n}]"
        if k == "householder":
            h = rng.standard_normal(n); h /= np.linalg.norm(h)
            return lo.opHouseholder(T(h, dev)), D(np.eye(n) - 2 * np.outer(h, h)), True, True, f"{k}[{m}x{n}]"
        if k == "hermitian":
            A, d = rng.standard_normal((n, n)), rng.standard_normal(n)
            L = np.tril(A, -1)
            return lo.opHermitian(T(d, dev), TM(A, dev)), D(L + L.T + np.diag(d)), True, True, f"{k}[{m}x{n}]"
        if k in ("lbfgs", "invlbfgs", "lsr1"):
            ctor = {"lbfgs": lo.LBFGSOperator, "invlbfgs": lo.InverseLBFGSOperator, "lsr1": lo.LSR1Operator}[k]
            op = ctor(n, mem=3, device=dev)
            for _ in range(4):
                s = rng.uniform(-1, 1, n)
                lo.push(op, T(s, dev), T(s * rng.uniform(0.5, 2.0, n) + 1e-2 * rng.standard_normal(n), dev))
            Mq = lo.Matrix(op).cpu().numpy()                       # dense image of the operator itself (symmetric)
            if np.isfinite(Mq).all() and np.abs(Mq).max() < 1e8:    # random pairs in tiny dimensions can make the
                return op, D(Mq), True, True, f"{k}[{m}x{n}]"       # secant updates blow up (in the reference too)
            A = rng.standard_normal((m, n))
            return lo.LinearOperatorFromMatrix(TM(A, dev)), D(A), True, True, f"dense[{m}x{n}]"
        if k == "blockdiag":
            cut = int(rng.integers(1, n)) if n > 1 else 1
            if n == 1:
                d = rng.standard_normal(1)
                return lo.BlockDiagonalOperator(lo.opDiagonal(T(d, dev))), D(np.diag(d)), True, True, f"{k}[{m}x{n}]"
            d1, A2 = rng.standard_normal(cut), rng.standard_normal((n - cut, n - cut))
            sparse2 = bool(rng.integers(2))                        # second block dense or sparse (both in the one-launch table)
            if sparse2:
                A2 = A2 * (rng.random(A2.shape) < 0.4)
            M = np.zeros((n, n)); M[:cut, :cut] = np.diag(d1); M[cut:, cut:] = A2
            return (lo.BlockDiagonalOperator(lo.opDiagonal(T(d1, dev)), self.csc(A2) if sparse2 else TM(A2, dev)), D(M), True, True,
                    f"{k}{'+sparse' if sparse2 else ''}[{m}x{n}]")
        if k == "kron":
            A, B = rng.standard_normal((2, 2)), rng.standard_normal((m // 2, n // 2))
            return lo.kron(TM(A, dev), TM(B, dev)), D(np.kron(A, B)), True, True, f"{k}[{m}x{n}]"
        if k == "restriction":
            idx = rng.choice(n, size=m, replace=False) + 1
            R = lo.opRestriction(idx.tolist(), n, S=S)
            return R, D(np.eye(n)[idx - 1, :]), False, False, f"{k}[{m}x{n}]"
        idx = rng.choice(m, size=n, replace=False) + 1           # extension: (m, n) with m > n
        E = lo.opExtension(idx.tolist(), m, S=S)
        return E, D(np.eye(m)[:, idx - 1]), False, False, f"{k}[{m}x{n}]"

    def honouring(self, m, n, depth, transposed):
        """A subtree that honours (α, β) in prod (or tprod when `transposed`): needed wherever the reference hands
        an operand β = 1 to accumulate (second summand, second hcat block, second block of a transposed vcat)."""
        for _ in range(20):
            t = self.tree(m, n, depth)
            if t[3 if transposed else 2]:
                return t
        A = self.rng.standard_normal((m, n))
        return self.lo.LinearOperatorFromMatrix(TM(A, self.dev)), D(A), True, True, f"dense[{m}x{n}]"

    def honouring_both(self, m, n, depth):
        for _ in range(20):
            t = self.tree(m, n, depth)
            if t[2] and t[3]:
                return t
        A = self.rng.standard_normal((m, n))
        return self.lo.LinearOperatorFromMatrix(TM(A, self.dev)), D(A), True, True, f"dense[{m}x{n}]"

    def tree(self, m, n, depth):
        """(op, dense, hN, hT, description), following how each combinator forwards α, β (operations.jl:102-215,
        cat.jl:7-33,65-91, adjtrans.jl): a sum applies its second operand with β = 1 in both directions (and `a - b`
        with -α), `x*op` applies op with x*α, hcat accumulates its second block in prod, vcat in tprod; a product hands the caller's α, β to its LEFT factor
        in prod and to its RIGHT factor in tprod; op[rows, cols] = R*op*E never honours them."""
        lo, rng = self.lo, self.rng
        if depth == 0:
            return self.leaf(m, n)
        c = rng.integers(8)
        if c in (0, 1):
            a, A, aN, aT, da = self.tree(m, n, depth - 1)
            b, B, bN, bT, db = self.honouring_both(m, n, depth - 1)   # the second summand accumulates (β = 1)
            if c == 0:
                return a + b, A + B, aN and bN, aT and bT, f"({da} + {db})"
            return a - b, A - B, aN and bN, aT and bT, f"({da} - {db})"
        if c == 2:
            k = int(rng.integers(1, 9))
            a, A, aN, aT, da = self.tree(m, k, depth - 1); b, B, bN, bT, db = self.tree(k, n, depth - 1)
            return a * b, A @ B, aN, bT, f"({da} * {db})"
        if c == 3:
            x = float(rng.uniform(-2, 2))
            a, A, aN, aT, d = self.honouring_both(m, n, depth - 1)     # x*op applies op with α scaled by x
            return x * a, A.scaled(x), aN, aT, f"({x:.2f} * {d})"
        if c == 4:
            a, A, aN, aT, d = self.tree(n, m, depth - 1)
            t = int(rng.integers(2))
            return (a.T if t else a.H), A.T, aT, aN, f"{d}{'.T' if t else '.H'}"
        if c == 5 and n >= 2:
            k = int(rng.integers(1, n))
            a, A, aN, aT, da = self.tree(m, k, depth - 1)
            b, B, bN, bT, db = self.honouring(m, n - k, depth - 1, transposed=False)
            return lo.hcat(a, b), hstack(A, B), aN and bN, aT and bT, f"hcat({da}, {db})"
        if c == 6 and m >= 2:
            k = int(rng.integers(1, m))
            a, A, aN, aT, da = self.tree(k, n, depth - 1)
            b, B, bN, bT, db = self.honouring(m - k, n, depth - 1, transposed=True)
            return lo.vcat(a, b), vstack(A, B), aN and bN, aT and bT, f"vcat({da}, {db})"
        if c == 7:
            a, A, _, _, d = self.tree(m + 2, n + 3, depth - 1)
            rows = (rng.choice(m + 2, size=m, replace=False) + 1).tolist()
            cols = (rng.choice(n + 3, size=n, replace=False) + 1).tolist()
            return a[rows, cols], A.pick(np.array(rows) - 1, np.array(cols) - 1), False, False, f"{d}[{rows},{cols}]"
        return self.leaf(m, n)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("MXLO_FUZZ_SEEDS", "300"))))
def test_random_operator_tree_vs_dense(lo, dev, seed):
    g = Gen(lo, dev, seed)
    rng = g.rng
    hi = 13 if seed % 4 else 40
    m, n = int(rng.integers(1, hi)), int(rng.integers(1, hi))
    op, DM, honours, honours_t, desc = g.tree(m, n, depth=int(rng.integers(1, 4)))
    M = DM.M
    assert op.shape == M.shape == (m, n)
    v, w, r0 = rng.standard_normal(n), rng.standard_normal(m), rng.standard_normal(m)
    scale = max(np.linalg.norm(DM.G, 2), 1e-300)           # forward-error scale of the tree's arithmetic

    def close(got, want, vec, extra=0.0):
        return np.linalg.norm(got - want) <= 1e-10 * (scale * np.linalg.norm(vec) + extra)

    assert close((op * T(v, dev)).cpu().numpy(), M @ v, v), desc
    assert close((op.T * T(w, dev)).cpu().numpy(), M.T @ w, w), desc
    assert close((op.H * T(w, dev)).cpu().numpy(), M.T @ w, w), desc
    res = T(r0.copy(), dev)
    lo.mul(res, op, T(v, dev), 3.0, -4.0)
    if honours:
        assert close(res.cpu().numpy(), 3.0 * (M @ v) - 4.0 * r0, 3 * v, 4 * np.linalg.norm(r0)), desc
    if honours_t:                                            # the same through the transpose
        rt0 = rng.standard_normal(n)
        rt = T(rt0.copy(), dev)
        lo.mul(rt, op.T, T(w, dev), 3.0, -4.0)
        assert close(rt.cpu().numpy(), 3.0 * (M.T @ w) - 4.0 * rt0, 3 * w, 4 * np.linalg.norm(rt0)), desc
    assert np.linalg.norm(lo.Matrix(op).cpu().numpy() - M) <= 1e-10 * scale * max(m, n), desc


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("MXLO_BDFUZZ_SEEDS", "80"))))
def test_random_blockdiagonal_vs_dense(lo, dev, seed):
    """Single-launch BlockDiagonalOperator over random mixes of diagonal / dense / identity / zero blocks of random
    sizes (every 16-byte phase, tile boundaries falling anywhere, rectangular dense and zero blocks), both dtypes,
    N / T / C, (α, β) incl. β = 0 on NaN-filled output, against the dense assembly."""
    rng = np.random.default_rng(7000 + seed)
    dtype = torch.float64 if seed % 2 == 0 else torch.float32
    npd = np.float64 if dtype == torch.float64 else np.float32
    S = lo.Storage(dtype, dev)
    nblk = int(rng.integers(1, 40))
    big = rng.integers(4) == 0
    ops, dense = [], []
    for _ in range(nblk):
        k = rng.integers(4)
        if k == 0:
            m = int(rng.integers(1, 5000 if big else 70))
            d = rng.standard_normal(m).astype(npd)
            ops.append(lo.opDiagonal(T(d, dev))); dense.append(np.diag(d.astype(np.float64)))
        elif k == 1:
            m, n = int(rng.integers(1, 40)), int(rng.integers(1, 40))
            A = rng.standard_normal((m, n)).astype(npd)
            ops.append(TM(A, dev) if rng.integers(2) else lo.LinearOperatorFromMatrix(TM(A, dev)))
            dense.append(A.astype(np.float64))
        elif k == 2:
            m = int(rng.integers(1, 3000 if big else 50))
            ops.append(lo.opEye(dtype, m, S=S)); dense.append(np.eye(m))
        else:
            m, n = int(rng.integers(1, 30)), int(rng.integers(1, 30))
            ops.append(lo.opZeros(dtype, m, n, S=S)); dense.append(np.zeros((m, n)))
    nr, nc = sum(a.shape[0] for a in dense), sum(a.shape[1] for a in dense)
    BD = lo.BlockDiagonalOperator(*ops)
    assert hasattr(BD, "_keepalive") and BD.shape == (nr, nc)        # the fused single-launch path

    def ref(x, transposed):
        out, r, c = np.zeros(nc if transposed else nr), 0, 0
        for a in dense:
            if transposed:
                out[c:c + a.shape[1]] = a.T @ x[r:r + a.shape[0]]
            else:
                out[r:r + a.shape[0]] = a @ x[c:c + a.shape[1]]
            r += a.shape[0]; c += a.shape[1]
        return out

    tol = 1e-12 if dtype == torch.float64 else 3e-5
    x, xt = rng.standard_normal(nc).astype(npd), rng.standard_normal(nr).astype(npd)
    for op, vec, tr in ((BD, x, False), (BD.T, xt, True), (BD.H, xt, True)):
        want = ref(vec.astype(np.float64), tr)
        out = torch.full((want.size,), float("nan"), dtype=dtype, device=dev)
        lo.mul(out, op, T(vec, dev), 1.0, 0.0)
        assert np.linalg.norm(out.cpu().numpy() - want) <= tol * (np.linalg.norm(want) + np.linalg.norm(vec)), (seed, tr)
        r0 = rng.standard_normal(want.size).astype(npd)
        out = T(r0.copy(), dev)
        lo.mul(out, op, T(vec, dev), 2.0, -3.0)
        want2 = 2.0 * want - 3.0 * r0
        assert np.linalg.norm(out.cpu().numpy() - want2) <= tol * (np.linalg.norm(want2) + 2 * np.linalg.norm(vec) + 3 * np.linalg.norm(r0)), (seed, tr)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("MXLO_IDXFUZZ_SEEDS", "80"))))
def test_random_restriction_extension_bit_exact(lo, dev, seed):
    """opRestriction / opExtension over random UnitRanges, StepRanges (positive and NEGATIVE steps), index vectors with
    duplicates (last write wins, special-operators.jl:171-174), strictly increasing sets of random density (index plans)
    and scalars; data of 4 and 8 bytes; views at odd offsets.
    Pure data movement: bit-exact against NumPy indexing."""
    rng = np.random.default_rng(9000 + seed)
    dtype = [torch.float64, torch.float32, torch.int64, torch.int32][seed % 4]
    ncol = int(rng.integers(1, 5000))
    off = int(rng.integers(0, 4))
    base = torch.arange(1, ncol + off + 1, device=dev).to(dtype) * (3 if dtype.is_floating_point else 1)
    v = base[off:off + ncol]                                   # a view at any 4/8-byte phase
    vh = v.cpu().numpy()
    S = lo.Storage(dtype, dev)
    kind = rng.integers(5)
    if kind == 4:                                               # round 5: a strictly increasing set of random density —
        dens = [0.9, 0.5, 0.2, 0.05][int(rng.integers(4))]      # bit mask + ranks from 1/32 (extension) / 1/8 (restriction)
        idx0 = np.flatnonzero(rng.random(ncol) < dens)
        I = (idx0 + 1).tolist()
    elif kind == 0:
        a = int(rng.integers(1, ncol + 1)); b = int(rng.integers(a, ncol + 1))
        I, idx0 = lo.jrange(a, b), np.arange(a, b + 1) - 1
    elif kind == 1:
        step = int(rng.integers(1, 9)) * (1 if rng.integers(2) else -1)
        a = int(rng.integers(1, ncol + 1)); b = int(rng.integers(1, ncol + 1))
        if (b - a) * step < 0:
            a, b = b, a
        I = lo.jrange(a, b, step)
        idx0 = np.arange(a, b + (1 if step > 0 else -1), step) - 1
    elif kind == 2:
        k = int(rng.integers(0, 2 * ncol))
        idx0 = rng.integers(0, ncol, k)                         # duplicates likely
        I = (idx0 + 1).tolist()
    else:
        idx0 = np.array([int(rng.integers(0, ncol))])
        I = int(idx0[0] + 1)
    R = lo.opRestriction(I, ncol, S=S)
    assert R.shape == (len(idx0), ncol)
    got = (R * v).cpu().numpy()
    assert np.array_equal(got, vh[idx0]), (seed, kind)
    u = (torch.arange(1, len(idx0) + 1, device=dev).to(dtype) * 7)
    want = np.zeros(ncol, dtype=vh.dtype)
    want[idx0] = u.cpu().numpy()                                # NumPy fancy assignment: last write wins, like the loop
    for E in (R.T, lo.opExtension(I, ncol, S=S)):
        ext = E * u
        assert np.array_equal(ext.cpu().numpy(), want), (seed, kind)
