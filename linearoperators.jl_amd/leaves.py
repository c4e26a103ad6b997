"""Leaf operators: same constructor names and flags as the reference, closures that call libmxlo.so.

Mirrors src/special-operators.jl (opEye, opOnes, opZeros, opDiagonal, opRestriction, opExtension,
BlockDiagonalOperator), src/linalg.jl:77-127 (opHouseholder, opHermitian), src/kron.jl and the
dense-matrix constructor src/constructors.jl:19-29.

Index arguments keep Julia's 1-BASED convention (the C ABI receives them exactly as Julia stores
them): ``opRestriction([1, 2, 4, 7], 10)``; ``jrange(3, 6)`` is Julia's ``3:6`` (inclusive),
``jrange(1, 7, 2)`` is ``1:2:7``; ``slice(None)`` / ``Ellipsis`` stand for ``:``.
"""
from __future__ import annotations

import operator
import os

import ctypes as C
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib
from .device import Storage, check_vec, ctx_of, dtype_code, get_ctx, indexed_device, ptr, storage_of
from .operators import (AbstractLinearOperator, LinearOperator, LinearOperatorException, _c4, adjoint, columnwise, compose,
                        conj_scalar, issymmetric, ishermitian, mul, scalar_flags, state_version, storage_type, to_dense, transpose)


def _default_device() -> torch.device:
    return torch.device("cuda", torch.cuda.current_device())


def _S(T: torch.dtype, S: Optional[Storage]) -> Storage:
    return S if S is not None else Storage(T, _default_device())


# ----------------------------------------------------------------------------- opEye / opOnes / opZeros
def mulOpEye(res, v, alpha, beta, n_min):
    """mulOpEye! — src/special-operators.jl:36-44 (tail gets `β` itself when β != 0)."""
    ctx = get_ctx(res.device)
    if res.dtype.is_complex:
        _lib.call("mxlo_eye_mul_c", ctx.handle, dtype_code(res.dtype, True), ptr(res), ptr(v), n_min, res.numel(),
                  *_c4(alpha, beta), scalar_flags(res.dtype, alpha, beta) | _lib.TAIL_BETA)
        return
    _lib.call("mxlo_eye_mul", ctx.handle, dtype_code(res.dtype), ptr(res), ptr(v), n_min, res.numel(),
              float(alpha), float(beta), scalar_flags(res.dtype, alpha, beta) | _lib.TAIL_BETA)


def opEye(T=torch.float64, nrow: Optional[int] = None, ncol: Optional[int] = None, S: Optional[Storage] = None):
    """opEye(T, n; S) / opEye(T, nrow, ncol; S) — src/special-operators.jl:46-73; opEye() is the size-less identity (:5-34)."""
    if nrow is None and not isinstance(T, int):
        from .operators import UniversalEye
        return UniversalEye()
    if isinstance(T, int):               # opEye(n) / opEye(nrow, ncol): T defaults to Float64
        T, nrow, ncol = torch.float64, T, nrow
    n = nrow
    S = _S(T, S)
    if ncol is None or nrow == ncol:
        prod = columnwise(lambda res, v, a, b: mulOpEye(res, v, a, b, n))
        op = LinearOperator(T, n, n, True, True, prod, prod, prod, S=S)
        op._leaf = ("eye", n, n)
        op._deps = ()
        return op
    n_min = min(nrow, ncol)
    prod = columnwise(lambda res, v, a, b: mulOpEye(res, v, a, b, n_min))
    op = LinearOperator(T, nrow, ncol, False, False, prod, prod, prod, S=S)
    op._leaf = ("eye", nrow, ncol)
    op._deps = ()
    return op


def mulOpOnes(res, v, alpha, beta):
    """mulOpOnes! — src/special-operators.jl:79-85."""
    ctx = get_ctx(res.device)
    _lib.call("mxlo_ones_mul", ctx.handle, dtype_code(res.dtype), ptr(res), res.numel(), ptr(v), v.numel(),
              float(alpha), float(beta), scalar_flags(res.dtype, alpha, beta))


def opOnes(T=torch.float64, nrow: Optional[int] = None, ncol: Optional[int] = None, S: Optional[Storage] = None):
    """opOnes(T, nrow, ncol; S) — src/special-operators.jl:87-101."""
    if isinstance(T, int):
        T, nrow, ncol = torch.float64, T, nrow
    prod = lambda res, v, a, b: mulOpOnes(res, v, a, b)
    op = LinearOperator(T, nrow, ncol, nrow == ncol, nrow == ncol, prod, prod, prod, S=_S(T, S))
    op._deps = ()
    return op


def mulOpZeros(res, v, alpha, beta):
    """mulOpZeros! — src/special-operators.jl:102-108."""
    ctx = get_ctx(res.device)
    if res.dtype.is_complex:
        b = complex(beta)
        _lib.call("mxlo_zeros_mul_c", ctx.handle, dtype_code(res.dtype, True), ptr(res), res.numel(), b.real, b.imag,
                  scalar_flags(res.dtype, 0, beta))
        return
    _lib.call("mxlo_zeros_mul", ctx.handle, dtype_code(res.dtype), ptr(res), res.numel(), float(beta),
              scalar_flags(res.dtype, 0, beta))


def opZeros(T=torch.float64, nrow: Optional[int] = None, ncol: Optional[int] = None, S: Optional[Storage] = None):
    """opZeros(T, nrow, ncol; S) — src/special-operators.jl:110-123."""
    if isinstance(T, int):
        T, nrow, ncol = torch.float64, T, nrow
    prod = columnwise(lambda res, v, a, b: mulOpZeros(res, v, a, b))
    op = LinearOperator(T, nrow, ncol, nrow == ncol, nrow == ncol, prod, prod, prod, S=_S(T, S))
    op._leaf = ("zeros", nrow, ncol)
    op._deps = ()
    return op


# ----------------------------------------------------------------------------- opDiagonal
def mulSquareOpDiagonal(res, d, v, alpha, beta, conj_d=False):
    """mulSquareOpDiagonal! — src/special-operators.jl:125-131 (`conj_d`: the ctprod! closure passes conj.(d), :139-141)."""
    n = res.numel()
    if res.dtype is torch.float64 and (d.numel() != 1 or n == 1):
        _lib.call("mxlo_diag_mul", ctx_of(res).handle, _lib.F64, res.data_ptr(), d.data_ptr(), v.data_ptr(), n, n,
                  float(alpha), float(beta), 0)
        return
    ctx = get_ctx(res.device)
    if res.dtype.is_complex:
        _lib.call("mxlo_diag_mul_c", ctx.handle, dtype_code(res.dtype, True), ptr(res), ptr(d), ptr(v), n, n,
                  *_c4(alpha, beta), scalar_flags(res.dtype, alpha, beta) | (_lib.CONJ_D if conj_d else 0))
        return
    flags = scalar_flags(res.dtype, alpha, beta) | (_lib.D_SCALAR if d.numel() == 1 and n != 1 else 0)
    _lib.call("mxlo_diag_mul", ctx.handle, dtype_code(res.dtype), ptr(res), ptr(d), ptr(v), n, n, float(alpha),
              float(beta), flags)


def mulOpDiagonal(res, d, v, alpha, beta, n_min, conj_d=False):
    """mulOpDiagonal! — src/special-operators.jl:144-151 (tail zeroed regardless of β)."""
    ctx = get_ctx(res.device)
    if res.dtype.is_complex:
        _lib.call("mxlo_diag_mul_c", ctx.handle, dtype_code(res.dtype, True), ptr(res), ptr(d), ptr(v), n_min,
                  res.numel(), *_c4(alpha, beta), scalar_flags(res.dtype, alpha, beta) | (_lib.CONJ_D if conj_d else 0))
        return
    _lib.call("mxlo_diag_mul", ctx.handle, dtype_code(res.dtype), ptr(res), ptr(d), ptr(v), n_min, res.numel(),
              float(alpha), float(beta), scalar_flags(res.dtype, alpha, beta))


def opDiagonal(*args):
    """opDiagonal(d) / opDiagonal(nrow, ncol, d) — src/special-operators.jl:133-165: symmetric = true,
    hermitian = isreal(d), ctprod! multiplies by conj.(d)."""
    if len(args) == 1:
        d = check_vec(args[0], "d")
        dtype_code(d.dtype, True)
        n = d.numel()
        prod = columnwise(lambda res, v, a, b: mulSquareOpDiagonal(res, d, v, a, b))
        if d.dtype.is_complex:
            ctprod = columnwise(lambda res, w, a, b: mulSquareOpDiagonal(res, d, w, a, b, conj_d=True))   # conj.(d) (:140)
            op = LinearOperator(d.dtype, n, n, True, False, prod, prod, ctprod, S=storage_of(d))  # isreal(d): a
        else:                                                                                     # type property
            op = LinearOperator(d.dtype, n, n, True, True, prod, prod, prod, S=storage_of(d))
            op._leaf = ("diag", d)
        op._deps = (d,)
        return op
    nrow, ncol, d = args
    d = check_vec(d, "d")
    dtype_code(d.dtype, True)
    if nrow == ncol <= d.numel():
        return opDiagonal(d[:nrow].clone())       # d[1:nrow] copies in Julia (:157)
    n_min = min(nrow, ncol)
    if d.numel() < n_min:
        raise LinearOperatorException("shape mismatch")
    prod = columnwise(lambda res, v, a, b: mulOpDiagonal(res, d, v, a, b, n_min))
    ctprod = columnwise(lambda res, w, a, b: mulOpDiagonal(res, d, w, a, b, n_min, conj_d=True)) if d.dtype.is_complex else prod
    op = LinearOperator(d.dtype, nrow, ncol, False, False, prod, prod, ctprod, S=storage_of(d))
    op._deps = (d,)
    return op


# ----------------------------------------------------------------------------- restriction / extension
class jrange:
    """Julia range ``start:step:stop`` (inclusive, 1-based): UnitRange when step == 1."""

    def __init__(self, start: int, stop: int, step: int = 1):
        if step == 0:
            raise ValueError("step cannot be zero")
        self.start, self.step = int(start), int(step)
        self.len = max(0, (int(stop) - int(start)) // int(step) + 1)

    def __len__(self):
        return self.len

    def to_numpy(self):
        return self.start + self.step * np.arange(self.len, dtype=np.int64)


def _is_colon(I) -> bool:
    return I is Ellipsis or (isinstance(I, slice) and I == slice(None))


def sorted_scatter_plan(idx_h: np.ndarray):
    """`res .= 0; res[I] = u` (src/special-operators.jl:171-174) is sequential: with duplicates the LAST write wins.
    Resolved once into a SORTED plan — strictly increasing target indices + the position in u of the surviving write —
    so that the apply is the segment-owner kernel (res written exactly once, in full vectors) whatever the order of I.
    Returns (sorted unique indices int64, source positions int64) or (idx, None) when I is already strictly increasing."""
    idx_h = np.asarray(idx_h, dtype=np.int64).reshape(-1)
    nrow = idx_h.size
    svals, last_rev = np.unique(idx_h[::-1], return_index=True)
    spos = (nrow - 1 - last_rev).astype(np.int64)
    if spos.size == nrow and (nrow == 0 or bool(np.all(spos == np.arange(nrow)))):
        return idx_h, None
    return svals.astype(np.int64), spos


PLAN_MIN_DENSITY = 1          # 0 switches the mask + rank form off (tests compare the two device implementations)
PLAN_MIN_DENSITY_INV = 32     # ... used when nidx >= n / 32: below, the index list is the smaller description
PLAN_GATHER_DENSITY_INV = 8   # ... and for the restriction (res = v[I]) when nidx >= n / 8


class _IndexPlan:
    """RAII wrapper of `mxlo_index_plan` (strictly increasing indices as bit mask + ranks, built from host indices)."""

    def __init__(self, ctx, sorted_idx_host: np.ndarray, n: int):
        import ctypes as C
        import weakref
        self.handle = C.c_void_p()
        _lib.call("mxlo_index_plan_create", ctx.handle, sorted_idx_host.ctypes.data, sorted_idx_host.size, int(n), C.byref(self.handle))
        self._ctx = ctx                                  # the plan refers to its ctx: keep it alive
        weakref.finalize(self, _lib.lib().mxlo_index_plan_destroy, self.handle)


def opRestriction(Idx, ncol: int, S: Optional[Storage] = None, device=None):
    """opRestriction(I, ncol; S) — src/special-operators.jl:176-203. The operator's eltype is the
    index integer type (Int64), like the reference (`LinearOperator{I, Vector{I}}`, :193)."""
    dev = indexed_device(device) if device is not None else (indexed_device(S.device) if S is not None else _default_device())
    if _is_colon(Idx):
        return opEye(torch.int64, ncol, S=Storage(torch.int64, dev))    # :201
    if isinstance(Idx, (int, np.integer)):
        Idx = [int(Idx)]                                                 # :203
    storage = S if S is not None else Storage(torch.int64, dev)
    if isinstance(Idx, jrange):
        start, step, ln = Idx.start, Idx.step, Idx.len
        last = start + (ln - 1) * step
        if ln > 0 and not (1 <= start <= ncol and 1 <= last <= ncol):
            raise LinearOperatorException(f"indices should be between 1 and {ncol}")

        def prod(res, v, a, b):   # mulRestrict! :167-169 (α, β ignored)
            ctx = get_ctx(res.device)
            _lib.call("mxlo_gather_range", ctx.handle, res.element_size(), ptr(res), ptr(v), v.numel(), start, step, ln)

        def tprod(res, u, a, b):  # multRestrict! :171-174
            ctx = get_ctx(res.device)
            _lib.call("mxlo_scatter_zero_range", ctx.handle, res.element_size(), ptr(res), res.numel(), ptr(u), start,
                      step, ln)
        nrow = ln
    else:
        idx_h = np.asarray(Idx.cpu() if isinstance(Idx, torch.Tensor) else Idx, dtype=np.int64).reshape(-1)
        if idx_h.size and not (idx_h.min() >= 1 and idx_h.max() <= ncol):
            raise LinearOperatorException(f"indices should be between 1 and {ncol}")
        nrow = idx_h.size
        idx_d = torch.from_numpy(idx_h.copy()).to(dev)
        svals, spos = sorted_scatter_plan(idx_h)
        sidx_d = idx_d if spos is None else torch.from_numpy(svals).to(dev)
        spos_d = None if spos is None else torch.from_numpy(spos).to(dev)

        # a dense enough index set is kept as bit mask + ranks (include/mxlo.h "index plans"): the applies stream the
        # long vector and never read the index list. The plan is per device; built here, once.
        plan = None
        if PLAN_MIN_DENSITY > 0 and svals.size > 0 and svals.size * PLAN_MIN_DENSITY_INV >= ncol:
            plan = _IndexPlan(get_ctx(dev), np.ascontiguousarray(svals, dtype=np.int64), ncol)
        # res = v[I] in I's own order: only when I is increasing — and from 1/8 density on (measured crossover with the
        # index-list gather, profiles/r05_bench_index.txt; the extension wins at every density above 1/32)
        gather_plan = plan if (spos is None and svals.size * PLAN_GATHER_DENSITY_INV >= ncol) else None

        def prod(res, v, a, b):
            ctx = get_ctx(res.device)
            if gather_plan is not None and res.device == dev:
                _lib.call("mxlo_gather_plan", ctx.handle, res.element_size(), ptr(res), ptr(v), v.numel(), gather_plan.handle)
                return
            _lib.call("mxlo_gather", ctx.handle, res.element_size(), ptr(res), ptr(v), v.numel(), ptr(idx_d), nrow)

        def tprod(res, u, a, b):
            ctx = get_ctx(res.device)
            if plan is not None and res.device == dev:
                _lib.call("mxlo_scatter_zero_plan", ctx.handle, res.element_size(), ptr(res), res.numel(), ptr(u), ptr(spos_d),
                          plan.handle)
                return
            _lib.call("mxlo_scatter_zero_sorted", ctx.handle, res.element_size(), ptr(res), res.numel(), ptr(u),
                      ptr(sidx_d), ptr(spos_d), sidx_d.numel())
    op = LinearOperator(torch.int64, nrow, ncol, False, False, prod, tprod, tprod, S=storage)
    op._deps = ()          # the index set is copied at construction
    return op


def opExtension(Idx, ncol: int, S: Optional[Storage] = None, device=None):
    """opExtension(I, ncol; S) = opRestriction(I, ncol; S)' — src/special-operators.jl:217-222."""
    if _is_colon(Idx):
        return opRestriction(Idx, ncol, S=S, device=device)
    return adjoint(opRestriction(Idx, ncol, S=S, device=device))


# ----------------------------------------------------------------------------- Householder / Hermitian
def mulHouseholder(res, h, v, alpha, beta):
    """mulHouseholder! — src/linalg.jl:77-83 (complex h: LinearAlgebra.dot conjugates it)."""
    if res.dtype is torch.float64:                      # the launch-bound case: nothing to decide, no flags
        _lib.call("mxlo_householder_mul", ctx_of(res).handle, _lib.F64, res.data_ptr(), h.data_ptr(), v.data_ptr(),
                  res.numel(), float(alpha), float(beta), 0)
        return
    ctx = get_ctx(res.device)
    if res.dtype.is_complex:
        _lib.call("mxlo_householder_mul_c", ctx.handle, dtype_code(res.dtype, True), ptr(res), ptr(h), ptr(v),
                  res.numel(), *_c4(alpha, beta), scalar_flags(res.dtype, alpha, beta))
        return
    _lib.call("mxlo_householder_mul", ctx.handle, dtype_code(res.dtype), ptr(res), ptr(h), ptr(v), res.numel(),
              float(alpha), float(beta), scalar_flags(res.dtype, alpha, beta))


def opHouseholder(h: torch.Tensor):
    """opHouseholder(h) — src/linalg.jl:85-95: symmetric=isreal(h), hermitian=true, tprod!=nothing,
    ctprod!=prod!. (The reference hard-codes S=Vector{T}; here S follows h.)"""
    h = check_vec(h, "h")
    dtype_code(h.dtype, True)
    n = h.numel()
    prod = lambda res, v, a, b: mulHouseholder(res, h, v, a, b)
    op = LinearOperator(h.dtype, n, n, not h.dtype.is_complex, True, prod, None, prod, S=storage_of(h))
    op._deps = (h,)
    return op


def mulHermitian(res, d, A, v, alpha, beta):
    """mulHermitian! — src/linalg.jl:97-103 with L = tril(A,-1) taken from A in the kernel."""
    ctx = get_ctx(res.device)
    n = res.numel()
    if res.dtype.is_complex:
        fl = scalar_flags(res.dtype, alpha, beta) | (0 if d.dtype.is_complex else _lib.D_REAL)
        _lib.call("mxlo_hermitian_mul_c", ctx.handle, dtype_code(res.dtype, True), ptr(res), ptr(d), ptr(A), A.stride(1),
                  ptr(v), n, *_c4(alpha, beta), fl)
        return
    _lib.call("mxlo_hermitian_mul", ctx.handle, dtype_code(res.dtype), ptr(res), ptr(d), ptr(A), A.stride(1),
              ptr(v), n, float(alpha), float(beta), scalar_flags(res.dtype, alpha, beta))


def _colmajor(M: torch.Tensor) -> torch.Tensor:
    """A Julia Matrix is column-major: return a 2-D tensor whose memory is column-major (stride(0) == 1).
    Column-major inputs are aliased. A row-major torch matrix is transposed-copied ONCE here — used only by opHermitian,
    whose kernel reads the strict LOWER triangle of a column-major matrix (a row-major A would need the upper-triangle
    twin): build it from `A.t().contiguous().t()`-style column-major storage when later in-place updates of A must be
    seen. LinearOperator(M) and kron alias row-major inputs (N/T swapped)."""
    if M.dim() != 2:
        raise ValueError("matrix expected")
    if M.stride(0) == 1 and M.stride(1) >= max(1, M.shape[0]):
        return M
    return M.t().contiguous().t()


def opHermitian(*args):
    """opHermitian(d, A) / opHermitian(A) — src/linalg.jl:105-127. Complex A (ComplexF64 / ComplexF32): `L'` is the
    conjugate transpose, symmetric = isreal(A) = false, hermitian = true; d may be real (the reference test passes
    real.(diag(A)), test/test_linop.jl:362) or complex."""
    if hasattr(args[-1], "tocsc") and not isinstance(args[-1], torch.Tensor):      # a scipy.sparse matrix, as LinearOperator(M),
        Mc = args[-1].tocsc()                                                       # cat and BlockDiagonalOperator accept it
        Mc.sort_indices()
        dv = args[0].device if len(args) == 2 else None
        args = (*args[:-1], sparse_csc(Mc.indptr, Mc.indices, Mc.data, Mc.shape[0], Mc.shape[1], index_base=0, device=dv))
    if getattr(args[-1], "layout", None) in _SPARSE_LAYOUTS:
        return _opHermitian_sparse(*args)
    if len(args) == 1:
        A = _colmajor(args[0])
        d = torch.diagonal(A).clone()
    else:
        d, A = args
        A = _colmajor(A)
        d = check_vec(d, "d")
    m, n = A.shape
    if not (m == n == d.numel()):
        raise LinearOperatorException("shape mismatch")
    U = torch.promote_types(d.dtype, A.dtype)
    if U.is_complex:
        dtype_code(U, True)
        comp = torch.float64 if U == torch.complex128 else torch.float32
        if A.dtype != U:
            A = _colmajor(A.to(U))
        if d.dtype.is_complex:
            d = d if d.dtype == U else d.to(U)
        else:
            d = d if d.dtype == comp else d.to(comp)      # a real diagonal stays real: d .* v is Real * Complex
        # mul! on MATRICES is an extension here, for real and complex data alike (see the real branch below): column by column
        prod = columnwise(lambda res, v, a, b: mulHermitian(res, d, A, v, a, b))
        op = LinearOperator(U, m, m, False, True, prod, None, None, S=Storage(U, A.device))
        op._deps = (d, A)
        return op
    if d.dtype != U:
        d = d.to(U)
    if A.dtype != U:
        A = _colmajor(A.to(U))
    dtype_code(U)
    # mul! on MATRICES: a deliberate EXTENSION of the reference, not a match. Its closure ends in `(...)[:]`
    # (src/linalg.jl:99-101), which flattens an n x k product to length n*k, so `mul!(res::Matrix, opHermitian(d, A), V)`
    # with k > 1 throws DimensionMismatch upstream. Block Krylov callers want it, so here a matrix is applied column by
    # column — one mulHermitian per column, or (real data on the device) the block entry point, which reads the triangle
    # once per 4 columns and gives every column the bits of the single apply.
    prod = columnwise(lambda res, v, a, b: mulHermitian(res, d, A, v, a, b))

    def herm_block(res, V, a, b):
        if res.dtype != U or V.dtype != U:
            raise TypeError(f"mul! on matrices: {res.dtype} / {V.dtype} operands next to a {U} operator")
        ctx = get_ctx(res.device)
        _lib.call("mxlo_hermitian_mul_block", ctx.handle, dtype_code(U), ptr(res), _ld(res), ptr(d), ptr(A), A.stride(1), ptr(V), _ld(V),
                  m, V.shape[1], float(a), float(b), scalar_flags(res.dtype, a, b))
    prod._matrix = herm_block
    op = LinearOperator(U, m, m, True, True, prod, None, None, S=Storage(U, A.device))
    op._deps = (d, A)
    return op


# ----------------------------------------------------------------------------- dense matrix operator
def _opHermitian_sparse(*args):
    """opHermitian(d, A) / opHermitian(A) with a SPARSE A (src/linalg.jl:105-127 takes any AbstractMatrix; `tril(A, -1)`
    of a SparseMatrixCSC is sparse): L = the strict lower triangle, taken ONCE at construction like the reference's
    `tril` copy, as a sparse leaf; an apply is `res = α d.*v + β res` (diagonal leaf), then `res += α L v` and
    `res += α Lᵀ v` on the same handle (modes N and T of `mxlo_csc_mul`) — three launches, L read twice (12 B per
    stored entry each)."""
    A = args[-1]
    if A.layout != torch.sparse_coo:
        A = A.to_sparse_coo()
    A = A.coalesce()
    m, n = A.shape
    if A.dtype.is_complex:
        raise TypeError("sparse opHermitian: complex element types are not instantiated on the device path")
    rows, cols = A.indices()
    vals = A.values()
    if len(args) == 1:
        d = torch.zeros(m, dtype=A.dtype, device=vals.device)
        on = rows == cols
        d[rows[on]] = vals[on]
    else:
        d = check_vec(args[0], "d")
    if not (m == n == d.numel()):
        raise LinearOperatorException("shape mismatch")
    U = torch.promote_types(d.dtype, A.dtype)
    dtype_code(U)
    low = rows > cols
    order = torch.argsort(cols[low] * m + rows[low])                         # column-major order of the kept entries
    lr, lc, lv = rows[low][order], cols[low][order], vals[low][order].to(U).contiguous()
    ccol = torch.zeros(n + 1, dtype=torch.int64, device=vals.device)
    ccol[1:] = torch.cumsum(torch.bincount(lc, minlength=n), 0)
    L = LinearOperatorFromSparse(_csc_tensor(ccol, lr, lv, (m, n)))
    D = opDiagonal(d if d.dtype == U else d.to(U))
    Lt = transpose(L)

    def prod(res, v, a, b):                        # mulHermitian! (src/linalg.jl:97-103)
        mul(res, D, v, a, b)
        mul(res, L, v, a, 1.0)
        mul(res, Lt, v, a, 1.0)

    op = LinearOperator(U, m, m, True, True, prod, None, None, S=Storage(U, vals.device))
    op._deps = (d, lv)
    return op


def _stored_colmajor(M: torch.Tensor):
    """(column-major view of the SAME memory, transposed?) — a row-major torch matrix (torch's default layout) IS the
    column-major storage of its transpose, so it is aliased with the roles of N and T swapped instead of being copied
    (the reference closure aliases M, src/constructors.jl:19-29: later in-place updates of M must be seen). Anything
    else (non-unit strides both ways) is copied once to column-major."""
    if M.dim() != 2:
        raise ValueError("matrix expected")
    if M.stride(0) == 1 and M.stride(1) >= max(1, M.shape[0]):
        return M, False
    if M.stride(1) == 1 and M.stride(0) >= max(1, M.shape[1]):
        return M.t(), True
    return M.t().contiguous().t(), False


_SPARSE_LAYOUTS = (torch.sparse_csc, torch.sparse_csr, torch.sparse_coo)
_version_of = operator.attrgetter("_version")


def _csc_tensor(ccol, rows, vals, size, csr=False):
    """torch.sparse_csc_tensor / sparse_csr_tensor without torch's "beta state" UserWarning (the tensor is only a carrier
    of the three arrays here: no torch sparse kernel is ever run on it)."""
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", UserWarning)
        return (torch.sparse_csr_tensor if csr else torch.sparse_csc_tensor)(ccol, rows, vals, size=size)


class _CscHandle:
    """RAII wrapper of ``mxlo_csc`` (include/mxlo.h) plus the tensors it keeps referenced."""

    def __init__(self, ctx, dtype, m, n, colptr, rowval, nzval, index_base):
        self.ctx, self.keep = ctx, (colptr, rowval, nzval)
        self.h = C.c_void_p()
        _lib.call("mxlo_csc_create", ctx.handle, dtype_code(dtype, dtype.is_complex), m, n, ptr(colptr), ptr(rowval), ptr(nzval), index_base,
                  C.byref(self.h))

    def info(self):
        a = (C.c_int64 * 8)()
        _lib.call("mxlo_csc_info", self.h, a)
        return {"m": a[0], "n": a[1], "nnz": a[2], "chunks_n": a[3], "chunks_t": a[4], "long_rows": a[5], "long_cols": a[6],
                "chunk": a[7]}

    def __del__(self):
        try:
            _lib.lib().mxlo_csc_destroy(self.h)
        except Exception:
            pass


def sparse_csc(colptr, rowval, nzval, m: int, n: int, index_base: int = 1, device=None) -> torch.Tensor:
    """A device `torch.sparse_csc` tensor from the three arrays of a Julia `SparseMatrixCSC` (`index_base` 1, as Julia
    stores them) or 0-based ones; host arrays are uploaded, `nzval` that already lives on the device is aliased."""
    dev = torch.device(device) if device is not None else (nzval.device if isinstance(nzval, torch.Tensor) and nzval.is_cuda
                                                           else _default_device())
    t = lambda a, dt=None: (a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))).to(device=dev, dtype=dt)
    cp, rv = t(colptr, torch.int64) - index_base, t(rowval, torch.int64) - index_base
    return _csc_tensor(cp, rv, t(nzval), (m, n))


def LinearOperatorFromSparse(M: torch.Tensor, symmetric: bool = False, hermitian: bool = False,
                             S: Optional[Storage] = None):
    """LinearOperator(M::SparseMatrixCSC) — src/constructors.jl:15-29: prod!/tprod!/ctprod! = `mul!(res, M, v, α, β)`,
    `mul!(res, transpose(M), …)`, `mul!(res, adjoint(M), …)`, which the SparseArrays stdlib implements as a column sweep
    (A*x) and a per-column gather (Aᵀ*x). Device form: `mxlo_csc_*` (include/mxlo.h) — a row gather-reduce on a
    compressed-row view in both modes, f64 accumulation in fixed order.

    `M` is a `torch.sparse_csc` tensor (torch's 0-based int64 indices: exactly the three arrays of a SparseMatrixCSC
    shifted by one — see `sparse_csc`) or `torch.sparse_csr` (the CSC storage of Mᵀ: aliased with N and T swapped, like a
    row-major dense matrix); COO is converted once. The VALUES are aliased: in-place updates of `M.values()` are seen
    (Aᵀ*x reads them directly; A*x re-gathers its row-ordered snapshot when the tensor's version counter moved —
    `lo.touched(M.values())` after writes torch cannot see). The sparsity pattern is fixed at construction."""
    if M.layout == torch.sparse_coo:
        M = M.coalesce().to_sparse_csc()
    tr = M.layout == torch.sparse_csr
    if M.dim() != 2 or M.layout not in (torch.sparse_csc, torch.sparse_csr):
        raise ValueError("LinearOperator(M): a 2-d sparse_csc / sparse_csr / sparse_coo tensor expected")
    nrow, ncol = M.shape
    vals = M.values()
    T = vals.dtype
    if T.is_complex and os.environ.get("MXLO_SPARSE_COMPLEX_PLANES", "0") == "1":
        return _sparse_complex(M, symmetric, hermitian, S)       # the real-planes form: the tests' independent device implementation
    dtype_code(T, T.is_complex)
    if tr:                                   # CSR of M == CSC of transpose(M)
        cp, rv, sm, sn = M.crow_indices(), M.col_indices(), ncol, nrow
    else:
        cp, rv, sm, sn = M.ccol_indices(), M.row_indices(), nrow, ncol
    if not vals.is_cuda:
        raise RuntimeError(f"operands must live on a GPU, got {vals.device}: there is no CPU fallback")
    cp, rv = cp.to(torch.int64).contiguous(), rv.to(torch.int64).contiguous()
    if not vals.is_contiguous():
        raise ValueError("sparse LinearOperator(M): M.values() must be contiguous")
    ctx = get_ctx(vals.device)
    handle = _CscHandle(ctx, T, sm, sn, cp, rv, vals, 0)
    seen = [state_version(vals)]
    fwd, bwd = (_lib.OP_T, _lib.OP_N) if tr else (_lib.OP_N, _lib.OP_T)
    if T.is_complex:
        # stored = M (CSC) or transpose(M) (CSR alias): M*v, transpose(M)*u run on the values as stored; M'*w = conj(Mᵀ)*w.
        # The kernel's mode C conjugates the stored values in the TRANSPOSED sweep, so it is M' only for CSC storage; a CSR
        # alias gets M' = conj(stored)*w through the conj sandwich  conj(stored * conj(w))  (three launches).
        code = dtype_code(T, True)

        def cspmv(res, v, a, b, mode):
            get_ctx(res.device)
            if res.dtype != T or v.dtype != T:
                raise TypeError(f"sparse {T} operator: {res.dtype} / {v.dtype} vectors")
            if mode == _lib.OP_N:
                tok = state_version(vals)
                if tok != seen[0]:
                    _lib.call("mxlo_csc_refresh", handle.h)
                    seen[0] = tok
            _lib.call("mxlo_csc_mul_c", handle.h, ptr(res), ptr(v), *_c4(a, b), mode, scalar_flags(res.dtype, a, b))

        def cadj_of_csr(res, w, a, b):                     # M' w with stored = transpose(M):  conj(stored * conj(w))
            from .operators import conj_into
            wc = torch.empty_like(w)
            conj_into(wc, w)
            tmp = torch.empty(res.shape[0], dtype=T, device=res.device)
            cspmv(tmp, wc, 1.0, 0.0, _lib.OP_N)
            conj_into(tmp, tmp)
            ctx = get_ctx(res.device)
            _lib.call("mxlo_eye_mul_c", ctx.handle, code, ptr(res), ptr(tmp), res.shape[0], res.shape[0], *_c4(a, b),
                      scalar_flags(res.dtype, a, b))

        prod = columnwise(lambda res, v, a, b: cspmv(res, v, a, b, fwd))
        tprod = columnwise(lambda res, u, a, b: cspmv(res, u, a, b, bwd))
        ctprod = columnwise(cadj_of_csr if tr else (lambda res, w, a, b: cspmv(res, w, a, b, _lib.OP_C)))
        op = LinearOperator(T, nrow, ncol, symmetric, hermitian, prod, tprod, ctprod,
                            S=S if S is not None else Storage(T, vals.device))
        op._sparse_src = M
        op._csc = handle
        op._deps = (vals,)
        return op

    def spmv(res, v, a, b, mode):
        get_ctx(res.device)
        if mode == _lib.OP_N:                # the row-ordered snapshot of the values
            tok = state_version(vals)
            if tok != seen[0]:
                _lib.call("mxlo_csc_refresh", handle.h)
                seen[0] = tok
        _lib.call("mxlo_csc_mul", handle.h, ptr(res), ptr(v), float(a), float(b), mode, scalar_flags(res.dtype, a, b))

    def spmm(res, m, a, b, mode):
        """res, m: column-major matrices — the stored matrix is read once per 8 columns (mxlo_csc_mul_block)"""
        get_ctx(res.device)
        if res.dtype != T or m.dtype != T:
            raise TypeError(f"mul! on matrices: {res.dtype} / {m.dtype} operands next to a sparse {T} matrix")
        if mode == _lib.OP_N:
            tok = state_version(vals)
            if tok != seen[0]:
                _lib.call("mxlo_csc_refresh", handle.h)
                seen[0] = tok
        _lib.call("mxlo_csc_mul_block", handle.h, ptr(res), _ld(res), ptr(m), _ld(m), m.shape[1], float(a), float(b), mode,
                  scalar_flags(res.dtype, a, b))

    prod = columnwise(lambda res, v, a, b: spmv(res, v, a, b, fwd))
    tprod = columnwise(lambda res, u, a, b: spmv(res, u, a, b, bwd))
    prod._matrix = lambda res, m, a, b: spmm(res, m, a, b, fwd)
    tprod._matrix = lambda res, m, a, b: spmm(res, m, a, b, bwd)
    op = LinearOperator(T, nrow, ncol, symmetric, hermitian, prod, tprod, tprod,
                        S=S if S is not None else Storage(T, vals.device))
    inf = handle.info()
    if not tr and inf["long_rows"] == 0 and inf["long_cols"] == 0:
        op._leaf = ("csc", handle, vals, seen)    # fused BlockDiagonalOperator: MXLO_BLK_CSC (no row / column beyond one chunk)
    op._sparse_src = M
    op._csc = handle
    op._deps = (vals,)
    return op


def _sparse_complex(M: torch.Tensor, symmetric: bool, hermitian: bool, S: Optional[Storage]):
    """Sparse M with a complex element type (test/test_linop.jl:44, test/test_cat.jl:5-25 build operators from
    `simple_sparse_matrix(ComplexF64, …)`): the device sparse kernels are instantiated for real element types, so M is
    split ONCE into two real sparse operands with the same pattern, Mr + i·Mi (a snapshot: `tril`-like copy semantics, unlike
    the aliased real path), and an apply runs on real planes:
        prod!    y = M x    : yr = Mr xr − Mi xi,   yi = Mr xi + Mi xr
        tprod!   y = Mᵀ x   : the same on the transposed sweeps
        ctprod!  y = Mᴴ x   : yr = Mrᵀ xr + Miᵀ xi, yi = Mrᵀ xi − Miᵀ xr
    four real sparse applies between one split and one join pass (`mxlo_split_c`, `mxlo_join_c` with the complex α, β)."""
    from .operators import LinearOperatorException
    T = M.values().dtype
    dtype_code(T, True)
    R = torch.float64 if T == torch.complex128 else torch.float32
    nrow, ncol = M.shape
    tr = M.layout == torch.sparse_csr
    cp, rv = (M.crow_indices(), M.col_indices()) if tr else (M.ccol_indices(), M.row_indices())
    mk = lambda v: _csc_tensor(cp, rv, v.contiguous(), (nrow, ncol), csr=tr)
    vals = M.values()
    planes_ = torch.view_as_real(vals.detach().clone())        # (.real on values() of a sparse tensor raises in torch 2.10)
    Lr, Li = LinearOperatorFromSparse(mk(planes_[:, 0])), LinearOperatorFromSparse(mk(planes_[:, 1]))
    dev = vals.device
    bufs = {}

    def planes(nin, nout):
        key = (nin, nout, torch.cuda.current_stream(dev).cuda_stream)
        if key not in bufs:
            if len(bufs) >= 4:
                bufs.clear()
            bufs[key] = tuple(torch.empty(k, dtype=R, device=dev) for k in (nin, nin, nout, nout))
        return bufs[key]

    def apply(res, v, a, b, wrap, sign):
        if res.dtype != T or v.dtype != T:
            raise TypeError(f"sparse {T} operator: {res.dtype} / {v.dtype} vectors")
        nin, nout = v.shape[0], res.shape[0]
        xr, xi, yr, yi = planes(nin, nout)
        ctx = get_ctx(dev)
        code = dtype_code(T, True)
        _lib.call("mxlo_split_c", ctx.handle, code, ptr(xr), ptr(xi), ptr(v), nin)
        Or, Oi = wrap(Lr), wrap(Li)
        mul(yr, Or, xr, 1.0, 0.0)
        mul(yr, Oi, xi, -sign, 1.0)
        mul(yi, Or, xi, 1.0, 0.0)
        mul(yi, Oi, xr, sign, 1.0)
        _lib.call("mxlo_join_c", ctx.handle, code, ptr(res), ptr(yr), ptr(yi), nout, *_c4(a, b), scalar_flags(res.dtype, a, b))

    ident = lambda o: o
    prod = columnwise(lambda res, v, a, b: apply(res, v, a, b, ident, 1.0))
    tprod = columnwise(lambda res, u, a, b: apply(res, u, a, b, transpose, 1.0))
    ctprod = columnwise(lambda res, w, a, b: apply(res, w, a, b, transpose, -1.0))
    op = LinearOperator(T, nrow, ncol, symmetric, hermitian, prod, tprod, ctprod, S=S if S is not None else Storage(T, dev))
    op._sparse_src = M
    op._deps = (vals,)
    return op


def LinearOperatorFromMatrix(M: torch.Tensor, symmetric: bool = False, hermitian: bool = False,
                             S: Optional[Storage] = None):
    """LinearOperator(M) — src/constructors.jl:15-29 (prod!/tprod!/ctprod! = gemv N/T/C). M is ALIASED, never copied,
    when it is column-major or row-major (a row-major M is the column-major storage of Mᵀ: N and T swap). A sparse M
    (torch.sparse_csc / _csr / _coo) goes to `LinearOperatorFromSparse`."""
    if not isinstance(M, torch.Tensor) and hasattr(M, "tocsc"):                              # a scipy.sparse matrix on the host
        Mc = M.tocsc()
        Mc.sort_indices()
        M = sparse_csc(Mc.indptr, Mc.indices, Mc.data, Mc.shape[0], Mc.shape[1], index_base=0,
                       device=S.device if S is not None else None)
    if M.layout in _SPARSE_LAYOUTS:
        return LinearOperatorFromSparse(M, symmetric, hermitian, S)
    nrow, ncol = M.shape
    St, tr = _stored_colmajor(M)
    cplx = St.dtype.is_complex
    dtype_code(St.dtype, True)
    sm, sn = St.shape
    ld = St.stride(1) if sn > 1 else max(1, sm)
    if cplx:
        # stored = M (tr False) or transpose(M) (tr True, row-major alias):  M*v, transpose(M)*u, M'*w
        fwd, bwd, cbwd = (_lib.OP_T, _lib.OP_N, _lib.OP_J) if tr else (_lib.OP_N, _lib.OP_T, _lib.OP_C)

        def gemv(res, v, a, b, mode):
            ctx = get_ctx(res.device)
            _lib.call("mxlo_gemv_c", ctx.handle, dtype_code(St.dtype, True), ptr(res), ptr(St), sm, sn, ld, ptr(v),
                      *_c4(a, b), mode, scalar_flags(res.dtype, a, b))
    else:
        fwd, bwd = (_lib.OP_T, _lib.OP_N) if tr else (_lib.OP_N, _lib.OP_T)
        cbwd = bwd                                                     # real element types: C == T

        def gemv(res, v, a, b, mode):
            ctx = get_ctx(res.device)
            _lib.call("mxlo_gemv", ctx.handle, dtype_code(St.dtype), ptr(res), ptr(St), sm, sn, ld, ptr(v), float(a),
                      float(b), mode, scalar_flags(res.dtype, a, b))

        def gemv_block(res, m, a, b, mode):
            """res, m: column-major matrices of St's dtype — one pass over the stored matrix per 8 columns"""
            ctx = get_ctx(res.device)
            k = m.shape[1]
            if res.dtype != St.dtype or m.dtype != St.dtype:
                raise TypeError(f"mul! on matrices: {res.dtype} / {m.dtype} operands next to a {St.dtype} matrix")
            _lib.call("mxlo_gemv_block", ctx.handle, dtype_code(St.dtype), ptr(res), _ld(res), ptr(St), sm, sn, ld,
                      ptr(m), _ld(m), k, float(a), float(b), mode, scalar_flags(res.dtype, a, b))

    # on matrices the reference closure `mul!(res, M, m, α, β)` is a GEMM: one GEMV per column here (same numbers)
    prod = columnwise(lambda res, v, a, b: gemv(res, v, a, b, fwd))
    tprod = columnwise(lambda res, u, a, b: gemv(res, u, a, b, bwd))
    ctprod = columnwise(lambda res, w, a, b: gemv(res, w, a, b, cbwd))
    if not cplx:                                  # real data: the whole block in one call (M read once per 8 columns)
        prod._matrix = lambda res, m, a, b: gemv_block(res, m, a, b, fwd)
        tprod._matrix = lambda res, m, a, b: gemv_block(res, m, a, b, bwd)
        ctprod._matrix = lambda res, m, a, b: gemv_block(res, m, a, b, cbwd)
    op = LinearOperator(St.dtype, nrow, ncol, symmetric, hermitian, prod, tprod, ctprod,
                        S=S if S is not None else Storage(St.dtype, St.device))
    if not tr and not cplx:
        op._leaf = ("dense", St, ld)                 # block-diagonal descriptor tables take column-major real blocks as they are
    op._dense_src = M
    op._deps = (M,)
    return op


# ----------------------------------------------------------------------------- BlockDiagonalOperator
class _BlockDiagHandle:
    def __init__(self, ctx, dtype, descs):
        self.ctx = ctx
        arr = (_lib.BlockDesc * len(descs))(*descs)
        self.h = C.c_void_p()
        _lib.call("mxlo_blockdiag_create", ctx.handle, dtype_code(dtype), arr, len(descs), C.byref(self.h))

    def __del__(self):
        try:
            _lib.lib().mxlo_blockdiag_destroy(self.h)
        except Exception:
            pass


def BlockDiagonalOperator(*ops, S: Optional[Storage] = None):
    """BlockDiagonalOperator(M1, ..., Mn; S) — src/special-operators.jl:249-294.

    When every block is a diagonal / dense-matrix / sparse-matrix / identity / zero block the whole operator is ONE
    device launch over a descriptor table (mxlo_blockdiag_mul); any other block type falls back to
    the reference's own structure (a host loop of inner `mul!` on views, :258-267) — still the HIP
    leaves, just one launch per block."""
    from .operators import _as_op, _promote_eltype, promote_storage
    ops = [_as_op(o) if isinstance(o, torch.Tensor) or hasattr(o, "tocsc") else o for o in ops]
    nrow = sum(o.size(1) for o in ops)
    ncol = sum(o.size(2) for o in ops)
    T = _promote_eltype(*ops)
    S = S if S is not None else promote_storage(*[storage_type(o) for o in ops])
    symm = all(issymmetric(o) for o in ops)
    herm = all(ishermitian(o) for o in ops)

    fusable = all(getattr(o, "_leaf", None) is not None and o.eltype == T for o in ops) and T.is_floating_point
    if fusable:
        descs, k, j = [], 0, 0
        keep, sparse_blocks = [], []
        for o in ops:
            leaf = o._leaf
            m, n = o.shape
            if leaf[0] == "diag":
                descs.append(_lib.BlockDesc(_lib.BLK_DIAG, 0, k, j, m, n, leaf[1].data_ptr(), 0))
                keep.append(leaf[1])
            elif leaf[0] == "dense":
                descs.append(_lib.BlockDesc(_lib.BLK_DENSE, 0, k, j, m, n, leaf[1].data_ptr(), leaf[2]))
                keep.append(leaf[1])
            elif leaf[0] == "eye" and m == n:
                descs.append(_lib.BlockDesc(_lib.BLK_EYE, 0, k, j, m, n, None, 0))
            elif leaf[0] == "zeros":
                descs.append(_lib.BlockDesc(_lib.BLK_ZEROS, 0, k, j, m, n, None, 0))
            elif leaf[0] == "csc":
                descs.append(_lib.BlockDesc(_lib.BLK_CSC, 0, k, j, m, n, leaf[1].h.value, 0))
                keep.append(leaf[1])
                sparse_blocks.append(leaf)
            else:
                fusable = False
                break
            k += m
            j += n
    if fusable:
        ctx = get_ctx(S.device)
        handle = _BlockDiagHandle(ctx, T, descs)
        sparse_vals = [leaf[2] for leaf in sparse_blocks]
        seen_sum = [-1]

        def bd(res, x, a, b, mode):
            get_ctx(res.device)
            if mode == _lib.OP_N and sparse_blocks:
                # sparse blocks: row-ordered value snapshots (see the leaf). One C-level pass over the version counters
                # (a Python loop over 1024 blocks would cost more than the launch); the per-block check only on a change
                vsum = sum(map(_version_of, sparse_vals))
                if vsum != seen_sum[0]:
                    seen_sum[0] = vsum
                    for _, hcsc, vals, seen in sparse_blocks:
                        tok = state_version(vals)
                        if tok != seen[0]:
                            _lib.call("mxlo_csc_refresh", hcsc.h)
                            seen[0] = tok
            _lib.call("mxlo_blockdiag_mul", handle.h, ptr(res), ptr(x), float(a), float(b), mode,
                      scalar_flags(res.dtype, a, b))

        prod = lambda y, x, a, b: bd(y, x, a, b, _lib.OP_N)
        tprod = lambda y, x, a, b: bd(y, x, a, b, _lib.OP_T)
        ctprod = lambda y, x, a, b: bd(y, x, a, b, _lib.OP_C)
        op = LinearOperator(T, nrow, ncol, symm, herm, prod, tprod, ctprod, S=S)
        op._keepalive = (handle, keep, ops)
        op._deps = tuple(ops)
        return op

    def prod(y, x, a, b):        # :258-267
        k = j = 0
        for o in ops:
            m, n = o.shape
            mul(y[k:k + m], o, x[j:j + n], a, b)
            k += m
            j += n

    def tprod(y, x, a, b):       # :269-278
        k = j = 0
        for o in ops:
            m, n = o.shape
            mul(y[k:k + n], transpose(o), x[j:j + m], a, b)
            k += n
            j += m

    def ctprod(y, x, a, b):      # :280-289
        k = j = 0
        for o in ops:
            m, n = o.shape
            mul(y[k:k + n], adjoint(o), x[j:j + m], a, b)
            k += n
            j += m

    op = LinearOperator(T, nrow, ncol, symm, herm, prod, tprod, ctprod, S=S)
    op._deps = tuple(ops)
    return op


# ----------------------------------------------------------------------------- kron
def kron(A, B, *, complex_form=None):
    """kron(A, B) — src/kron.jl:10-49: (A ⊗ B) x = vec(B X Aᵀ).

    `complex_form` (complex element types only; ignored for real factors): how a complex product is spelled in real
    MFMA GEMMs. `"gauss"` (default) = 3 real GEMMs per complex product (`mxlo_kron_mul_c3`); it is NORMWISE stable
    only — the error of each component of K·x is bounded relative to ‖K‖·‖x‖, so a real or imaginary part that is
    tiny next to the other one loses relative accuracy (the reference's own criterion, `test/test_kron.jl:35`:
    1e-12·‖K‖₁, is normwise and is met). `"4gemm"` = the textbook 4 real GEMMs (`mxlo_kron_mul_c`), componentwise
    accurate like the reference's complex arithmetic, ≈ 25 % slower at 1024². `None` reads the process default from
    `MXLO_KRON_GAUSS` (unset / "1" → "gauss", "0" → "4gemm").

    The reference rebuilds a composite operator and materialises it with `m` single-vector products on
    every apply. Here an apply is two MFMA GEMMs on dense device matrices: matrix factors are aliased in place
    (row-major ones through a transposition flag), operator factors are materialised with `Matrix(op)`
    (src/abstract.jl:282-292) and re-materialised whenever their state changed — see `_KronFactor`."""
    def diag_of(X):
        """(is_diagonal_like, d or None, n) for opDiagonal / square opEye leaves."""
        leaf = getattr(X, "_leaf", None) if not isinstance(X, torch.Tensor) else None
        if leaf is not None and leaf[0] == "diag":
            return True, leaf[1], leaf[1].numel()
        if leaf is not None and leaf[0] == "eye" and leaf[1] == leaf[2]:
            return True, None, leaf[1]
        return False, None, 0

    isdA, dA, mA = diag_of(A)
    isdB, dB, pB = diag_of(B)
    if isdA and isdB:
        # both factors diagonal (identity = no vector): the fused row/col index-decomposition kernel,
        # res[r + c*p] = α*(dB[r]*(x[r + c*p]*dA[c])) (+ β res) — one HBM pass, no GEMM (src/kron.jl:14-22)
        dts = [t.dtype for t in (dA, dB) if t is not None] or [A.eltype if A.eltype.is_floating_point else torch.float64]
        Td = dts[0] if len(dts) == 1 else torch.promote_types(dts[0], dts[1])
        dtype_code(Td)
        dev = next((t.device for t in (dA, dB) if t is not None), storage_type(A).device)
        conv = {}     # promoted copies of a lower-precision diagonal, refreshed when the source changed

        def dvec(d):
            if d is None or d.dtype == Td:
                return d                              # aliased: later updates of d are seen, like the reference
            tok = state_version(d)
            if conv.get(id(d), (None, None))[0] != tok:
                conv[id(d)] = (tok, d.to(Td))
            return conv[id(d)][1]

        def kd(res, x, a, b):
            ctx = get_ctx(res.device)
            _lib.call("mxlo_kron_diag_mul", ctx.handle, dtype_code(Td), ptr(res), ptr(dvec(dA)), mA, ptr(dvec(dB)), pB,
                      ptr(x), float(a), float(b), scalar_flags(res.dtype, a, b))

        op = LinearOperator(Td, mA * pB, mA * pB, True, True, kd, kd, kd, S=Storage(Td, dev))
        op._deps = (A, B)
        return op

    fA, fB = _KronFactor(A), _KronFactor(B)
    T = torch.promote_types(fA.dtype, fB.dtype)
    if T.is_complex:
        return _kron_complex(A, B, fA, fB, T, complex_form)
    dtype_code(T)
    fA.T, fB.T = T, T
    m, n = fA.shape
    p, q = fB.shape
    dev = fA.device
    work = torch.empty(max(q * m, p * n), dtype=T, device=dev)

    def km(res, x, a, b, trans):
        # prod!: kron(A, B);  tprod!/ctprod! (real eltypes): the same formula on the transposed factors
        # (src/kron.jl:24-40). The factors are read in place either way: a transposition is a flag of the GEMM
        # kernel's operand layout, not a copy.
        ctx = get_ctx(res.device)
        As, ta = fA.get()
        Bs, tb = fB.get()
        _lib.call("mxlo_kron_mul_ex", ctx.handle, dtype_code(T), ptr(res), ptr(As), As.shape[0], As.shape[1], _ld(As),
                  ta ^ trans, ptr(Bs), Bs.shape[0], Bs.shape[1], _ld(Bs), tb ^ trans, ptr(x), ptr(work), float(a),
                  float(b), scalar_flags(res.dtype, a, b))

    prod = lambda res, x, a, b: km(res, x, a, b, 0)
    tprod = lambda res, x, a, b: km(res, x, a, b, 1)
    op = LinearOperator(T, m * p, n * q, fA.symmetric and fB.symmetric, fA.hermitian and fB.hermitian, prod, tprod,
                        tprod, S=Storage(T, dev))
    op._deps = (A, B)
    return op


def _kron_complex(A, B, fA, fB, T, complex_form=None):
    """kron with at least one complex factor (test/test_kron.jl:3-8: Float64 A, ComplexF64 B). The apply works on real
    PLANES: a complex factor is split into (re, im) column-major planes once (refreshed when its state token changes),
    a real factor is aliased as it is with no imaginary plane; every complex product is then 4 (2) real MFMA GEMMs
    (`mxlo_kron_mul_c`). tprod! transposes both factors, ctprod! also conjugates them (src/kron.jl:24-40)."""
    dtype_code(T, True)
    R = torch.float64 if T == torch.complex128 else torch.float32
    m, n = fA.shape
    p, q = fB.shape
    dev = fA.device

    class Planes:
        def __init__(self, f):
            self.f, self.cache, self.token, self.sums = f, None, object(), {}
            if not f.dtype.is_complex:
                f.T = R

        def get(self):
            f = self.f
            if not f.dtype.is_complex:                  # real factor: aliased (either layout) / tracked by _KronFactor
                M, t = f.get()
                return M, None, t
            src = f.src if f.src is not None else f.op
            tok = state_version(src)
            if self.cache is None or tok is None or tok != self.token:
                D = (f.src if f.src is not None else to_dense(f.op)).to(T)
                self.cache = (D.real.t().contiguous().t(), D.imag.t().contiguous().t())     # column-major planes
                self.token = tok
                self.sums = {}
            return self.cache[0], self.cache[1], 0

        def sum_plane(self, sign):
            """re + sign*im (Gauss form, `mxlo_plane_sum`), cached with the planes: once per factor state and sign."""
            if not self.f.dtype.is_complex:
                return None
            re, im = self.cache                      # km() has just called get(): the planes are current
            out = self.sums.get(sign)
            if out is None:
                out = torch.empty(re.shape[1], re.shape[0], dtype=R, device=dev).t()        # contiguous column-major
                ctx = get_ctx(dev)
                _lib.call("mxlo_plane_sum", ctx.handle, dtype_code(T, True), ptr(out), ptr(re), ptr(im), re.shape[0],
                          re.shape[1], _ld(re), float(sign))
                self.sums[sign] = out
            return out

    pA, pB = Planes(fA), Planes(fB)
    # Gauss form (3 real GEMMs per complex product, `mxlo_kron_mul_c3`, normwise stable) by default; the 4-GEMM form
    # (`mxlo_kron_mul_c`, componentwise) by `complex_form="4gemm"` or MXLO_KRON_GAUSS=0 — the tests use it as the
    # independent device implementation. The workspace covers the forward and the transposed shapes of both forms.
    if complex_form is None:
        complex_form = "gauss" if os.environ.get("MXLO_KRON_GAUSS", "1") != "0" else "4gemm"
    if complex_form not in ("gauss", "4gemm"):
        raise ValueError(f'kron: complex_form must be "gauss" or "4gemm", got {complex_form!r}')
    gauss = complex_form == "gauss"
    wsz = _lib.lib().mxlo_kron_c3_work_size
    need = max(int(wsz(m, n, 0, p, q, 0)), int(wsz(m, n, 1, p, q, 1)),
               2 * (max(q * n, p * m) + max(m * q, n * p) + max(p * m, q * n)) + 24)
    work = torch.empty(need, dtype=R, device=dev)

    def km(res, x, a, b, trans, conj):
        ctx = get_ctx(res.device)
        Ar, Ai, ta = pA.get()
        Br, Bi, tb = pB.get()
        if gauss:
            sgn = -1.0 if conj else 1.0
            _lib.call("mxlo_kron_mul_c3", ctx.handle, dtype_code(T, True), ptr(res), ptr(Ar), ptr(Ai), ptr(pA.sum_plane(sgn)),
                      Ar.shape[0], Ar.shape[1], _ld(Ar), (ta ^ trans) | (conj << 1), ptr(Br), ptr(Bi), ptr(pB.sum_plane(sgn)),
                      Br.shape[0], Br.shape[1], _ld(Br), (tb ^ trans) | (conj << 1), ptr(x), ptr(work), *_c4(a, b),
                      scalar_flags(res.dtype, a, b))
            return
        _lib.call("mxlo_kron_mul_c", ctx.handle, dtype_code(T, True), ptr(res), ptr(Ar), ptr(Ai), Ar.shape[0], Ar.shape[1],
                  _ld(Ar), (ta ^ trans) | (conj << 1), ptr(Br), ptr(Bi), Br.shape[0], Br.shape[1], _ld(Br),
                  (tb ^ trans) | (conj << 1), ptr(x), ptr(work), *_c4(a, b), scalar_flags(res.dtype, a, b))

    prod = lambda res, x, a, b: km(res, x, a, b, 0, 0)
    tprod = lambda res, x, a, b: km(res, x, a, b, 1, 0)
    ctprod = lambda res, x, a, b: km(res, x, a, b, 1, 1)
    op = LinearOperator(T, m * p, n * q, fA.symmetric and fB.symmetric, fA.hermitian and fB.hermitian, prod, tprod,
                        ctprod, S=Storage(T, dev))
    op._deps = (A, B)
    op.complex_form = complex_form
    return op


def _ld(M: torch.Tensor) -> int:
    return M.stride(1) if M.shape[1] > 1 else max(1, M.shape[0])


class _KronFactor:
    """One factor of `kron`: hands `mxlo_kron_mul_ex` a column-major STORED matrix plus a transposition flag.

    * a matrix (or a dense `LinearOperator(M)`) is aliased, never copied: column-major storage is passed as is,
      row-major storage (torch's default) is its own transpose stored column-major -> flag 1. Later in-place
      updates of the caller's matrix are therefore seen, like in the reference, which keeps `A` itself
      (src/kron.jl:10-22). Only a dtype promotion or an exotic stride forces a converted copy, and that copy is
      refreshed whenever the source tensor's version counter moved.
    * any other operator is materialised with `Matrix(op)` (src/abstract.jl:282-292), which the reference does on
      EVERY apply (src/kron.jl:18). Here the dense image is cached and rebuilt when the operator's state token
      (`operators.state_version`: push!/reset! counters, tensor versions of every leaf underneath) changed; an
      operator whose state cannot be tracked is rebuilt on every apply, exactly like the reference."""

    def __init__(self, X):
        self.T = None
        if isinstance(X, torch.Tensor):
            if X.dim() != 2:
                raise ValueError("matrix expected")
            self.src, self.op = X, None
            self.symmetric = self.hermitian = False
        else:
            src = getattr(X, "_dense_src", None)          # LinearOperator(M): alias M itself (either layout)
            if src is not None:
                self.src, self.op = src, None
            else:
                self.src, self.op = None, X
            self.symmetric, self.hermitian = issymmetric(X), ishermitian(X)
        if self.src is not None:
            self.shape, self.dtype, self.device = tuple(self.src.shape), self.src.dtype, self.src.device
        else:
            self.shape = tuple(self.op.shape)
            dt = self.op.eltype
            self.dtype = dt if (dt.is_floating_point or dt.is_complex) else torch.float64
            self.device = storage_type(self.op).device
        self._cache = None
        self._token = object()      # never equal to a real token

    def get(self):
        if self.src is not None:
            M = self.src
            if M.dtype == self.T:
                m, n = M.shape
                if M.stride(0) == 1 and (n == 1 or M.stride(1) >= max(1, m)):
                    return M, 0                       # column-major as stored
                if M.stride(1) == 1 and (m == 1 or M.stride(0) >= max(1, n)):
                    return M.t(), 1                   # row-major = the transpose stored column-major
            tok = state_version(M)
            if self._cache is None or tok != self._token:
                self._cache = M.to(self.T).t().contiguous().t()      # column-major converted copy
                self._token = tok
            return self._cache, 0
        tok = state_version(self.op)
        if self._cache is None or tok is None or tok != self._token:
            D = to_dense(self.op)                     # column-major view of a fresh buffer
            self._cache = D if D.dtype == self.T else D.to(self.T).t().contiguous().t()
            self._token = tok
        return self._cache, 0


# ----------------------------------------------------------------------------- ShiftedOperator
class _ShiftedData:
    """ShiftedData{T, OpH} — src/shifted_operators.jl:4-13 (σ is mutable, like the reference)."""

    def __init__(self, H, sigma):
        if H.size(1) != H.size(2):
            raise ValueError("Operator H must be square.")           # DimensionMismatch (:8)
        self.H, self.sigma = H, sigma

    @property
    def σ(self):
        return self.sigma

    @σ.setter
    def σ(self, value):                                                # op.data.σ = ... (test_shifted_operator.jl:66-70)
        self.sigma = value


class ShiftedOperatorType(AbstractLinearOperator):
    """`ShiftedOperator(H, σ)` = H + σI — src/shifted_operators.jl:56-103. Its products are the inner
    operator's `mul!` followed by one `axpy!(α σ, x, y)` (:16-25), here `mxlo_eye_mul` with β = 1."""
    _has_args5 = True                                                  # has_args5 / isallocated5 (:92-94)
    fuse = True                      # quasi-Newton H: fold the axpy! into the apply (bit-identical, see tests)

    def __init__(self, H, sigma=0):
        self.eltype = H.eltype
        T = self.eltype
        self.data = _ShiftedData(H, self._convert(sigma))
        self.nrow = self.ncol = H.size(1)
        self.symmetric = issymmetric(H)
        self._herm0 = ishermitian(H) and self._isreal(self.data.sigma)   # the constructor's `op.hermitian` field (:82-83)
        self.nprod = self.ntprod = self.nctprod = 0
        data = self.data

        def axpy(y, x, c):                                             # y = y + c x, c already of the callers' product type
            ctx = get_ctx(y.device)
            if y.dtype.is_complex:
                c = complex(c)
                # axpy! converts the scalar to eltype(y) and runs in that arithmetic: no width flags; β = 1 is Real
                _lib.call("mxlo_eye_mul_c", ctx.handle, dtype_code(y.dtype, True), ptr(y), ptr(x), y.numel(), y.numel(),
                          c.real, c.imag, 1.0, 0.0, _lib.BETA_REAL)
            else:
                _lib.call("mxlo_eye_mul", ctx.handle, dtype_code(y.dtype), ptr(y), ptr(x), y.numel(), y.numel(), c, 1.0, 0)

        def shifted(y, x, a, b, op, conj_sigma=False):
            sig = self._convert(data.sigma)                            # σ is a mutable field of eltype T (:6,:73)
            if conj_sigma:
                sig = conj_scalar(sig)                                 # (:45)
            skip = sig == 0 or a == 0                                  # (:21)
            if not skip and ShiftedOperatorType.fuse and hasattr(data.H, "_pending_shift") and not conj_sigma:
                # H is a quasi-Newton operator: the axpy! rides in the combine pass of its apply
                # (mxlo_qn_mul_shifted) — same per-element roundings, one launch and 3 vector passes fewer.
                data.H._pending_shift = float(sig)
                try:
                    mul(y, op, x, a, b)
                finally:
                    data.H._pending_shift = None
                return y
            mul(y, op, x, a, b)                                        # y = α H x + β y        (:18)
            if not skip:
                # α*σ in the callers' types (σ is a T), then axpy! converts it to T and runs in T arithmetic
                if y.dtype == torch.complex128:
                    c = complex(a) * complex(sig)
                elif y.dtype == torch.complex64:
                    c = complex(np.complex64(complex(a) * complex(sig)))
                elif y.dtype == torch.float64:
                    c = float(a) * float(sig)
                elif isinstance(a, (np.float32,)):
                    c = float(np.float32(a) * np.float32(sig))
                else:
                    c = float(np.float32(float(a) * float(sig)))
                axpy(y, x, c)                                          # y = y + (α σ) x        (:22)
            return y

        self.prod = lambda y, x, a, b: shifted(y, x, a, b, data.H)
        self.tprod = lambda y, x, a, b: shifted(y, x, a, b, transpose(data.H))
        self.ctprod = lambda y, x, a, b: shifted(y, x, a, b, adjoint(data.H), conj_sigma=True)   # H' + conj(σ) I (:40-50)

    def _convert(self, sigma):
        """convert(T, σ) (:73): a complex σ next to a real H is an InexactError in the reference."""
        T = self.eltype
        if T.is_complex:
            z = complex(sigma)
            return complex(np.complex64(z)) if T == torch.complex64 else z
        if isinstance(sigma, complex) or (isinstance(sigma, np.generic) and np.iscomplexobj(sigma)):
            if complex(sigma).imag != 0:
                raise TypeError(f"InexactError: cannot convert the complex shift {sigma} to the real eltype {T}")
            sigma = complex(sigma).real
        return np.float32(sigma) if T == torch.float32 else float(sigma)

    @staticmethod
    def _isreal(sigma):
        return complex(sigma).imag == 0

    @property
    def hermitian(self):                                               # op.hermitian && isreal(op.data.σ) (:89)
        return self._herm0 and self._isreal(self.data.sigma)

    @property
    def S(self):                                                       # storage_type(op.data.H) (:96)
        return storage_type(self.data.H)

    def _state_version(self):
        v = state_version(self.data.H)
        return None if v is None else ("shift", v, complex(self.data.sigma))


def ShiftedOperator(H, sigma=0):
    return ShiftedOperatorType(H, sigma)
