"""Host-side mirror of the reference's operator API for the `mul!` hot path.

Mirrors, name for name and argument for argument (Python spelling: ``mul!`` -> :func:`mul`,
``push!`` -> ``push`` ...), the reference's

* operator ABI            src/abstract.jl:30-59,122-131,147-153,176-196,203-244,282-292
* 5-arg / 3-arg ``mul!``  src/operations.jl:3-48  (shape check, counters, ``prod3!``)
* lazy wrappers            src/adjtrans.jl:1-261
* combinators              src/operations.jl:99-234
* cat                      src/cat.jl:1-129

Only *host control flow* lives here; every vector touch is a libmxlo.so call made by a
leaf closure (see :mod:`leaves`, :mod:`qn`). Error behaviour follows the reference:
``LinearOperatorException("shape mismatch")`` before any data is touched.
"""
from __future__ import annotations

import inspect
from typing import Callable, Optional

import numpy as np
import torch

from . import _lib
from .device import Storage, check_vec, dtype_code, get_ctx, ptr, storage_of


class LinearOperatorException(Exception):
    """src/abstract.jl:16-18"""


# ----------------------------------------------------------------------------- scalars
def _is_f64_scalar(x) -> bool:
    """Julia `Float64` / `ComplexF64`: a Python float / complex or a float64 / complex128 NumPy/torch scalar.
    Python ints are Julia `Int` (never widen a Float32 product); float32 / complex64 scalars stay 32-bit."""
    if isinstance(x, bool) or isinstance(x, (int, np.integer)):
        return False
    if isinstance(x, (float, complex)):
        return True
    if isinstance(x, np.floating):
        return x.dtype == np.float64
    if isinstance(x, np.complexfloating):
        return x.dtype == np.complex128
    if isinstance(x, torch.Tensor):
        return x.dtype in (torch.float64, torch.complex128)
    return True


def _is_complex_scalar(x) -> bool:
    if isinstance(x, torch.Tensor):
        return x.dtype.is_complex
    return isinstance(x, (complex, np.complexfloating))


def scalar_flags(dtype: torch.dtype, alpha, beta) -> int:
    """MXLO_ALPHA_F64 / MXLO_BETA_F64: which caller scalars are Float64 next to Float32 (ComplexF32) data. Julia
    evaluates the α-term in promote_type(typeof(α), T) and the β-term in promote_type(typeof(β), T), each on its own
    (src/special-operators.jl:126-129; SURVEY §8a, mixed precision), so the two flags are independent. Complex
    data additionally gets MXLO_ALPHA_REAL / MXLO_BETA_REAL for Real scalars (Real*Complex is componentwise)."""
    fl = 0
    if dtype in (torch.float32, torch.complex64):
        fl |= (_lib.ALPHA_F64 if _is_f64_scalar(alpha) else 0) | (_lib.BETA_F64 if _is_f64_scalar(beta) else 0)
    if dtype.is_complex:
        fl |= (0 if _is_complex_scalar(alpha) else _lib.ALPHA_REAL) | (0 if _is_complex_scalar(beta) else _lib.BETA_REAL)
    return fl


def conj_scalar(x):
    """conj(α) of the wrapper routing (src/adjtrans.jl:131,134): the identity on Real scalars."""
    return x.conjugate() if _is_complex_scalar(x) else x


def _c4(alpha, beta):
    """(re, im, re, im) doubles of two caller scalars for the `_c` entry points."""
    a, b = complex(alpha), complex(beta)
    return a.real, a.imag, b.real, b.imag


def one(dtype: torch.dtype):
    """one(T): Float32 -> float32 scalar, Float64 -> Python float, ComplexF64 -> Python complex,
    ComplexF32 -> complex64 scalar, integer eltypes -> Python int."""
    if dtype.is_complex:
        return np.complex64(1) if dtype == torch.complex64 else complex(1.0)
    if not dtype.is_floating_point:
        return 1
    return np.float32(1) if dtype == torch.float32 else 1.0


def zero(dtype: torch.dtype):
    if dtype.is_complex:
        return np.complex64(0) if dtype == torch.complex64 else complex(0.0)
    if not dtype.is_floating_point:
        return 0
    return np.float32(0) if dtype == torch.float32 else 0.0


_NARGS_CACHE: dict = {}


def _nargs(f: Callable) -> int:
    """`hasmethod(op.prod!, (res, v, α, β))` stand-in (src/operations.jl:26): count positional params.
    Cached per code object: the reference pays this reflection on every `mul!`, the mirror only once."""
    try:
        return f._mxlo_nargs                           # cached on the function object itself (plain functions / lambdas)
    except AttributeError:
        pass
    code = getattr(f, "__code__", None)
    key = code if code is not None else id(f)
    n = _NARGS_CACHE.get(key)
    if n is not None:
        try:
            f._mxlo_nargs = n
        except (AttributeError, TypeError):            # builtins, bound methods, callables with __slots__
            pass
        return n
    try:
        sig = inspect.signature(f)
        n = 0
        for p in sig.parameters.values():
            if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD):
                n += 1
            elif p.kind == p.VAR_POSITIONAL:
                n = 4
                break
    except (TypeError, ValueError):
        n = 4
    _NARGS_CACHE[key] = n
    return n


def promote_storage(*ss: Optional[Storage]) -> Storage:
    """promote_type(storage_type(op1), storage_type(op2)) + isconcretetype check
    (src/operations.jl:137-147, src/cat.jl:47-49)."""
    ss = [s for s in ss if s is not None]
    dev = ss[0].device
    dt = ss[0].dtype
    for s in ss[1:]:
        if s.device != dev:
            raise LinearOperatorException(
                f"storage types {ss[0]} and {s} cannot be promoted to a concrete type. "
                "Ensure both operators use compatible storage types (e.g., both GPU or both CPU).")
        dt = torch.promote_types(dt, s.dtype)
    return Storage(dt, dev)


# ----------------------------------------------------------------------------- base types
class AbstractLinearOperator:
    """src/abstract.jl:30 — duck-typed on the fields nrow ncol symmetric hermitian prod tprod ctprod
    nprod ntprod nctprod (the structural contract quasi-Newton operators also satisfy)."""

    nrow: int
    ncol: int
    symmetric: bool
    hermitian: bool
    eltype: torch.dtype

    # --- size / flags (src/abstract.jl:203-244)
    @property
    def shape(self):
        return (self.nrow, self.ncol)

    def size(self, d: Optional[int] = None):
        if d is None:
            return self.shape
        if d == 1:
            return self.shape[0]
        if d == 2:
            return self.shape[1]
        raise LinearOperatorException("Linear operators only have 2 dimensions for now")

    # --- lazy wrappers (src/adjtrans.jl:33-45)
    @property
    def T(self):
        return transpose(self)

    @property
    def H(self):
        return adjoint(self)

    def conj(self):
        return conj(self)

    # --- arithmetic (src/operations.jl)
    def __neg__(self):
        return neg(self)

    def __pos__(self):
        return self

    def __mul__(self, other):
        if isinstance(other, AbstractLinearOperator):
            return compose(self, other)
        if isinstance(other, torch.Tensor) and other.dim() == 1:
            return apply(self, other)
        if isinstance(other, torch.Tensor) and other.dim() == 2:
            from .leaves import LinearOperatorFromMatrix
            return compose(self, LinearOperatorFromMatrix(other))
        if _is_number(other):
            return scale_op(self, other)
        return NotImplemented

    __matmul__ = __mul__

    def __rmul__(self, other):
        if _is_number(other):
            return scale_op(self, other)                     # src/operations.jl:181-183
        if isinstance(other, torch.Tensor) and other.dim() == 2:
            from .leaves import LinearOperatorFromMatrix
            return compose(LinearOperatorFromMatrix(other), self)
        return NotImplemented

    def __truediv__(self, x):
        if _is_number(x):
            return scale_op(self, one(self.eltype) / x)       # src/operations.jl:185
        return NotImplemented

    def __add__(self, other):
        if isinstance(other, AbstractLinearOperator):
            return add(self, other)
        if isinstance(other, torch.Tensor) and other.dim() == 2:
            from .leaves import LinearOperatorFromMatrix
            return add(self, LinearOperatorFromMatrix(other))
        if _is_number(other):                                 # src/operations.jl:222
            return add(self, scale_op(_ones_like(self), other))
        return NotImplemented

    def __radd__(self, other):
        if _is_number(other):                                 # src/operations.jl:223
            return add(scale_op(_ones_like(self), other), self)
        if isinstance(other, torch.Tensor) and other.dim() == 2:
            from .leaves import LinearOperatorFromMatrix
            return add(LinearOperatorFromMatrix(other), self)
        return NotImplemented

    def __sub__(self, other):
        if isinstance(other, AbstractLinearOperator):
            return add(self, neg(other))                      # src/operations.jl:226
        if _is_number(other):
            return self + (-other)                            # src/operations.jl:233
        if isinstance(other, torch.Tensor) and other.dim() == 2:
            from .leaves import LinearOperatorFromMatrix
            return add(self, neg(LinearOperatorFromMatrix(other)))
        return NotImplemented

    def __rsub__(self, other):
        if _is_number(other):
            return other + neg(self)                          # src/operations.jl:234
        return NotImplemented

    def __getitem__(self, key):                               # src/special-operators.jl:225-233
        from .leaves import opExtension, opRestriction
        rows, cols = key
        R = opRestriction(rows, self.size(1), S=None, device=storage_type(self).device)
        E = opExtension(cols, self.size(2), S=None, device=storage_type(self).device)
        return compose(compose(R, self), E)


def _is_number(x) -> bool:
    return isinstance(x, (int, float, complex, np.integer, np.floating, np.complexfloating)) and not isinstance(x, bool)


class LinearOperator(AbstractLinearOperator):
    """`LinearOperator{T,S}(nrow, ncol, symmetric, hermitian, prod!, tprod!, ctprod!)`
    (src/abstract.jl:38-51,122-131). `prod(res, v, alpha, beta)` is the 5-arg closure form,
    `prod(res, v)` the 3-arg form (handled by ``prod3``, with lazily allocated Mv/Mtu)."""

    def __init__(self, T: torch.dtype, nrow: int, ncol: int, symmetric: bool, hermitian: bool,
                 prod, tprod=None, ctprod=None, S: Optional[Storage] = None):
        self.eltype = T
        self.nrow, self.ncol = int(nrow), int(ncol)
        self.symmetric, self.hermitian = bool(symmetric), bool(hermitian)
        self.prod, self.tprod, self.ctprod = prod, tprod, ctprod
        self.nprod = self.ntprod = self.nctprod = 0
        if S is None:
            raise LinearOperatorException("storage type S is required (a device vector type)")
        self.S = S
        self.Mv = torch.empty(0, dtype=S.dtype)    # S(undef, 0)  src/abstract.jl:79-80 (no device touch yet)
        self.Mtu = torch.empty(0, dtype=S.dtype)

    def __repr__(self):            # src/abstract.jl:262-275
        return ("Linear operator\n  nrow: %d\n  ncol: %d\n  eltype: %s\n  symmetric: %s\n  hermitian: %s\n"
                "  nprod:   %d\n  ntprod:  %d\n  nctprod: %d\n" %
                (self.nrow, self.ncol, self.eltype, self.symmetric, self.hermitian, self.nprod, self.ntprod,
                 self.nctprod))


# counters (src/abstract.jl:147-153, src/adjtrans.jl:47-62)
def nprod(op):
    if isinstance(op, AdjointLinearOperator):
        return nctprod(op.parent)
    if isinstance(op, TransposeLinearOperator):
        return ntprod(op.parent)
    if isinstance(op, ConjugateLinearOperator):
        return nprod(op.parent)
    return op.nprod


def ntprod(op):
    if isinstance(op, (AdjointLinearOperator, TransposeLinearOperator)):
        return nprod(op.parent)
    if isinstance(op, ConjugateLinearOperator):
        return ntprod(op.parent)
    return op.ntprod


def nctprod(op):
    if isinstance(op, (AdjointLinearOperator, TransposeLinearOperator)):
        return nprod(op.parent)
    if isinstance(op, ConjugateLinearOperator):
        return nctprod(op.parent)
    return op.nctprod


def reset(op):
    """reset!(op) — src/abstract.jl:191-196 (quasi-Newton operators override, see qn.py)."""
    if hasattr(op, "_reset_data"):
        op._reset_data()
    op.nprod = op.ntprod = op.nctprod = 0
    return op


def issymmetric(op) -> bool:
    return op.parent.symmetric if isinstance(op, _Wrapper) else op.symmetric


def ishermitian(op) -> bool:
    return op.parent.hermitian if isinstance(op, _Wrapper) else op.hermitian


def has_args5(op) -> bool:          # src/abstract.jl:166
    if isinstance(op, _Wrapper):
        return has_args5(op.parent)
    if hasattr(op, "_has_args5"):
        return op._has_args5
    return _nargs(op.prod) == 4


def isallocated5(op) -> bool:       # src/abstract.jl:168
    if isinstance(op, _Wrapper):
        return isallocated5(op.parent)
    if hasattr(op, "_has_args5"):
        return True
    return not (op.Mv.numel() == 0 or op.Mtu.numel() == 0)


def storage_type(op) -> Storage:    # src/abstract.jl:176-184, src/adjtrans.jl:67-73
    if isinstance(op, _Wrapper):
        return storage_type(op.parent)
    if isinstance(op, torch.Tensor):
        return storage_of(op)
    return op.S


def eltype(op):
    return op.eltype


def size(op, d=None):
    return op.size(d)


def allocate_vectors_args3(op):     # src/operations.jl:3-8
    if isinstance(op, _Wrapper):
        return allocate_vectors_args3(op.parent)
    S = storage_type(op)
    op.Mv = S.undef(op.nrow)
    op.Mtu = op.Mv if op.nrow == op.ncol else S.undef(op.ncol)
    return op


# ----------------------------------------------------------------------------- device helpers for prod3!
def _axpby(res, Mv, alpha, beta):
    """res .= α .* Mv .+ β .* res (src/operations.jl:18) on the device."""
    ctx = get_ctx(res.device)
    n = res.numel()
    if res.dtype.is_complex:
        _lib.call("mxlo_eye_mul_c", ctx.handle, dtype_code(res.dtype, True), ptr(res), ptr(Mv), n, n,
                  *_c4(alpha, beta), scalar_flags(res.dtype, alpha, beta))
        return
    _lib.call("mxlo_eye_mul", ctx.handle, dtype_code(res.dtype), ptr(res), ptr(Mv), n, n, float(alpha),
              float(beta), scalar_flags(res.dtype, alpha, beta))


def _scale(res, alpha):
    """res .*= α (src/operations.jl:14)."""
    ctx = get_ctx(res.device)
    if res.dtype.is_complex:
        a = complex(alpha)
        _lib.call("mxlo_scale_c", ctx.handle, dtype_code(res.dtype, True), ptr(res), res.numel(), a.real, a.imag,
                  scalar_flags(res.dtype, alpha, 0))
        return
    _lib.call("mxlo_scale", ctx.handle, dtype_code(res.dtype), ptr(res), res.numel(), float(alpha),
              scalar_flags(res.dtype, alpha, 0))


def conj_into(res, v):
    """res .= conj.(v) (res may be v: conj!) — the identity, without a launch, for real element types."""
    if not v.dtype.is_complex:
        if res.data_ptr() != v.data_ptr():
            res.copy_(v)
        return res
    ctx = get_ctx(res.device)
    _lib.call("mxlo_conj_c", ctx.handle, dtype_code(v.dtype, True), ptr(res), ptr(v), v.numel())
    return res


# ----------------------------------------------------------------------------- operator state versions
def _probe_increment_version():
    """Pick the version-bump primitive ONCE, by observing its effect: `torch._C._increment_version` takes an iterable
    of tensors on current torch and a single tensor on older releases; the public wrapper
    `torch.autograd.graph.increment_version` is the last resort. A candidate is accepted only if the probe tensor's
    `_version` really advanced, so a signature change can never silently stop the bumps (kron planes and the
    factor-sum caches key on them)."""
    probe = torch.zeros(1)
    candidates = []
    raw = getattr(torch._C, "_increment_version", None)
    if raw is not None:
        candidates += [lambda t, raw=raw: raw((t,)), lambda t, raw=raw: raw(t)]
    pub = getattr(getattr(torch.autograd, "graph", None), "increment_version", None)
    if pub is not None:
        candidates.append(pub)
    for f in candidates:
        v0 = probe._version
        try:
            f(probe)
        except Exception:
            continue
        if probe._version > v0:
            return f
    raise ImportError("linearoperators_jl_amd: no working tensor version-bump primitive in this torch build "
                      "(torch._C._increment_version / torch.autograd.graph.increment_version)")


_increment_version = _probe_increment_version()


def touched(t: torch.Tensor):
    """libmxlo writes through raw pointers, which torch's in-place version counter does not see: bump it by hand so
    that operators built on `t` (and kron factors materialised from them) notice the change. Only inference
    tensors (which carry no version counter) are exempt; any other failure propagates."""
    try:
        _increment_version(t)
    except RuntimeError:
        if not t.is_inference():
            raise
    return t


def state_version(x):
    """A hashable token that changes whenever the action of `x` (an operator or a tensor) may have changed, or
    None when that cannot be known (an operator built from opaque user closures). `kron` re-materialises a
    non-matrix factor only when its token changed; with None it re-materialises on every apply, which is what the
    reference always does (src/kron.jl:14-22)."""
    if isinstance(x, torch.Tensor):
        return ("t", x.data_ptr(), x._version)
    if isinstance(x, _Wrapper):
        return state_version(x.parent)
    f = getattr(x, "_state_version", None)
    if callable(f):
        return f()
    deps = getattr(x, "_deps", None)
    if deps is None:
        return None
    vs = tuple(state_version(d) for d in deps)
    return None if any(v is None for v in vs) else vs


def prod3(res, prod, v, alpha, beta, Mv):
    """prod3! — src/operations.jl:10-20."""
    if beta == 0:
        prod(res, v)
        if alpha != 1:
            _scale(res, alpha)
    else:
        prod(Mv, v)
        _axpby(res, Mv, alpha, beta)


# ----------------------------------------------------------------------------- mul!
def mul(res: torch.Tensor, op, v: torch.Tensor, alpha=None, beta=None):
    """`mul!(res, op, v, α, β)` / `mul!(res, op, v)` — src/operations.jl:22-40, src/adjtrans.jl:90-261.

    3-arg form: α = one(T), β = zero(T) with T = eltype(v) (src/operations.jl:38-40)."""
    if alpha is None and beta is None:
        alpha, beta = one(v.dtype), zero(v.dtype)
    elif alpha is None or beta is None:
        raise TypeError("mul! takes either (res, op, v) or (res, op, v, alpha, beta)")
    if v.dim() == 2 or res.dim() == 2:
        return _mul_matrix(res, op, v, alpha, beta)
    vd, od = v.dtype, op.eltype
    if vd is not od:                                        # (the common case — same element type — skips all of this)
        if vd.is_complex and od.is_floating_point:
            return _mul_real_op_complex_vec(res, op, v, alpha, beta)
        if od.is_complex and vd.is_floating_point:
            v = _complex_of_real(v, od)                     # a real vector handed to a complex operator (test_adjtrans.jl:31-34)
    if not isinstance(op, _Wrapper):
        pass                                                # base operator: straight to the closure below
    elif isinstance(op, AdjointLinearOperator):
        return _mul_adjoint(res, op, v, alpha, beta)
    elif isinstance(op, TransposeLinearOperator):
        return _mul_transpose(res, op, v, alpha, beta)
    elif isinstance(op, ConjugateLinearOperator):
        # src/adjtrans.jl:226-249: mul!(res, p, conj.(v), α, β); conj!(res) — α, β and the incoming res are NOT
        # conjugated (reference behaviour, reproduced); real v skips the conj.(v) allocation (:238-249)
        vc = conj_into(torch.empty_like(v), v) if v.dtype.is_complex else v
        mul(res, op.parent, vc, alpha, beta)
        if res.dtype.is_complex:
            conj_into(res, res)
        return res
    if not (v.size(0) == op.ncol and res.size(0) == op.nrow):
        raise LinearOperatorException("shape mismatch")
    op.nprod += _COUNT                                      # increase_nprod!
    prod = op.prod
    if _nargs(prod) == 4:
        prod(res, v, alpha, beta)
    else:
        if not (beta == 0 or op.Mv.numel() != 0):
            allocate_vectors_args3(op)
        prod3(res, op.prod, v, alpha, beta, op.Mv)
    return touched(res)


def columnwise(fn):
    """Mark a device closure whose REFERENCE closure acts column by column when it is handed matrices (dense
    LinearOperator(M): `mul!(res, M, m, α, β)` is a GEMM; opDiagonal / opEye / opZeros: broadcasts). The device closures are
    vector kernels, so `_mul_matrix` applies such a closure to the columns one by one."""
    fn._columnwise = True
    return fn


def matrixwise(fn):
    """Mark a closure of this package that accepts matrices because its REFERENCE closure does: `-op`, `x*op`, `op1+op2`
    call `mul!(res::AbstractVecOrMat, …)` on their operands (src/operations.jl:103-105, 165-167, 187-197), and `mul`
    routes matrices. (`op1*op2` does not qualify: its vector temporaries fail on matrices in the reference too.)"""
    fn._matrixwise = True
    return fn


def _apply_closure_to_matrix(fn, res, m, alpha, beta):
    if fn is None:
        raise LinearOperatorException("Not implemented")                      # src/adjtrans.jl:155,223
    if getattr(fn, "_matrixwise", False):
        fn(res, m, alpha, beta)
        return touched(res)
    if not getattr(fn, "_columnwise", False):
        if (getattr(fn, "__module__", "") or "").startswith(__name__.rsplit(".", 1)[0]):
            raise LinearOperatorException("mul! on matrices: this operator's closure is defined on vectors only "
                                          "(dense LinearOperator(M), opDiagonal, opEye and opZeros take matrices)")
        fn(res, m, alpha, beta)                                               # a caller's closure gets the matrices as they are
        return touched(res)
    if res.dim() != 2 or m.dim() != 2 or res.shape[1] != m.shape[1]:
        raise LinearOperatorException("shape mismatch")
    def colmajor(X):                        # Julia layout (unit stride down a column, any leading dimension) or a copy of it
        ok = (X.shape[0] <= 1 or X.stride(0) == 1) and (X.shape[1] <= 1 or X.stride(1) >= max(1, X.shape[0]))
        return X if ok else X.t().contiguous().t()
    mc, rc = colmajor(m), colmajor(res)                                       # Julia layout: column j is contiguous
    block = getattr(fn, "_matrix", None)                                      # dense real M: one pass over M for the block
    if block is not None and m.shape[1] > 1 and mc.is_cuda:
        block(rc, mc, alpha, beta)
    else:
        for j in range(m.shape[1]):
            fn(rc[:, j], mc[:, j], alpha, beta)
    if rc is not res:
        res.copy_(rc)
    return touched(res)


def _mul_matrix(res, op, m, alpha, beta):
    """`mul!(res::AbstractMatrix, op, m::AbstractMatrix, α, β)` — src/operations.jl:34-36: the closure is handed the
    matrices (no shape check, no counter); wrappers: src/adjtrans.jl:139-156 (adjoint: hermitian parent, else ctprod!, else
    "Not implemented"), :207-224 (transpose: symmetric parent, else tprod!), :251-261 (conjugate: conj.(m), then conj!)."""
    if isinstance(op, (AdjointLinearOperator, TransposeLinearOperator)):
        p = op.parent
        if not (m.dim() == 2 and res.dim() == 2 and m.shape[0] == p.size(1) and res.shape[0] == p.size(2)
                and m.shape[1] == res.shape[1]):
            raise LinearOperatorException("shape mismatch")
        adj = isinstance(op, AdjointLinearOperator)
        if ishermitian(p) if adj else issymmetric(p):
            return _mul_matrix(res, p, m, alpha, beta)
        return _apply_closure_to_matrix(p.ctprod if adj else p.tprod, res, m, alpha, beta)
    if isinstance(op, ConjugateLinearOperator):
        _mul_matrix(res, op.parent, torch.conj_physical(m) if m.dtype.is_complex else m, alpha, beta)
        if res.dtype.is_complex:
            res.copy_(torch.conj_physical(res))
        return touched(res)
    if not hasattr(op, "prod"):
        raise LinearOperatorException("mul! on matrices: not defined for this operator type")
    # The reference hands the matrices to the closure unchecked (src/operations.jl:34-36) and BLAS / broadcast throw
    # DimensionMismatch; the device closures take RAW POINTERS and their sizes from the operator, so a wrongly shaped
    # operand would read or write out of bounds: check here, before anything is launched.
    if not (m.dim() == 2 and res.dim() == 2 and m.shape[0] == op.size(2) and res.shape[0] == op.size(1)
            and m.shape[1] == res.shape[1]):
        raise LinearOperatorException("shape mismatch")
    return _apply_closure_to_matrix(op.prod, res, m, alpha, beta)


def _ones_like(op):
    """`opOnes(op.nrow, op.ncol)` of `op ± x` (src/operations.jl:222-223) on the operator's device. The reference's opOnes is
    a Float64 operator whatever eltype(op) is; here it takes the REAL component type of eltype(op) (Float32 data stays
    Float32; next to a complex operator the sum promotes as Real + Complex does, test/test_linop.jl:164-176)."""
    from .leaves import opOnes
    st = storage_type(op)
    rt = {torch.complex128: torch.float64, torch.complex64: torch.float32}.get(op.eltype, op.eltype)
    return opOnes(rt, op.nrow, op.ncol, S=Storage(rt, st.device))


def _complex_of_real(v, ctype):
    """Complex{R} copy of a real vector (zero imaginary parts), formed on the device by the join kernel."""
    from . import _lib
    from .device import dtype_code, get_ctx, ptr
    comp = torch.float64 if ctype == torch.complex128 else torch.float32
    if v.dtype != comp:
        raise TypeError(f"a {v.dtype} vector next to a {ctype} operator: convert one of them")
    out = torch.empty(v.shape[0], dtype=ctype, device=v.device)
    ctx = get_ctx(v.device)
    _lib.call("mxlo_join_c", ctx.handle, dtype_code(ctype, True), ptr(out), ptr(v), None, v.shape[0], 1.0, 0.0, 0.0, 0.0,
              _lib.ALPHA_REAL | _lib.BETA_REAL)
    return out


_COUNT = 1      # what increase_nprod!/ntprod!/nctprod! add (0 while the second plane of a real-on-complex apply runs)


def _mul_real_op_complex_vec(res, op, v, alpha, beta):
    """A REAL operator (eltype Float64 / Float32; wrappers included: for a real operator adjoint == transpose) applied
    to complex vectors — `K * x` with `x::Vector{ComplexF64}` (test/test_kron.jl "issue110"). Julia runs the generic
    closure on the complex vectors; the device leaves are instantiated per element type, so the real operator is applied
    to the real and imaginary planes (two real 3-arg applies) and res = α*(yr + i*yi) (+ β*res) is formed in one pass.
    Equal to the reference up to rounding (1e-12-class), not bit-for-bit."""
    from . import _lib
    from .device import dtype_code, get_ctx, ptr
    if not res.dtype.is_complex or res.dtype != v.dtype:
        raise TypeError("a complex input vector needs a complex result vector of the same type")
    comp = torch.float64 if v.dtype == torch.complex128 else torch.float32
    if op.eltype != comp:
        raise TypeError(f"operator eltype {op.eltype} next to {v.dtype} vectors: convert one of them")
    if not (v.shape[0] == op.size(2) and res.shape[0] == op.size(1)):
        raise LinearOperatorException("shape mismatch")
    nin, nout = v.shape[0], res.shape[0]
    # plane temporaries live with the operator (lazily, like Mv / Mtu: operators are not re-entrant, src §5), keyed by
    # the stream they were last used on so that applies issued on two streams never share them
    root = op
    while isinstance(root, _Wrapper):
        root = root.parent
    key = (nin, nout, comp, v.device, torch.cuda.current_stream(v.device).cuda_stream if v.is_cuda else 0)
    cache = root.__dict__.setdefault("_planes", {})
    bufs = cache.get(key)
    if bufs is None:
        if len(cache) >= 4:
            cache.clear()
        bufs = cache[key] = tuple(torch.empty(k, dtype=comp, device=v.device) for k in (nin, nin, nout, nout))
    xr, xi, yr, yi = bufs
    ctx = get_ctx(v.device)
    code = dtype_code(v.dtype, True)
    _lib.call("mxlo_split_c", ctx.handle, code, ptr(xr), ptr(xi), ptr(v), nin)
    global _COUNT
    mul(yr, op, xr)
    saved, _COUNT = _COUNT, 0                           # the reference counts ONE product per apply (its closure runs once, on
    try:                                                # the complex vectors): the second plane is not a second mul!, for the
        mul(yi, op, xi)                                 # operator and for every operator its closure re-enters
    finally:
        _COUNT = saved
    _lib.call("mxlo_join_c", ctx.handle, code, ptr(res), ptr(yr), ptr(yi), nout, *_c4(alpha, beta),
              scalar_flags(res.dtype, alpha, beta))
    return touched(res)


def _call_t(res, f, v, alpha, beta, p):
    if _nargs(f) == 4:
        f(res, v, alpha, beta)
    else:
        if not (beta == 0 or p.Mtu.numel() != 0):
            allocate_vectors_args3(p)
        prod3(res, f, v, alpha, beta, p.Mtu)
    return touched(res)


def _call_conj_sandwich(res, f, v, alpha, beta, p):
    """conj!(res); f(res, conj.(v), conj(α), conj(β)); conj!(res) — src/adjtrans.jl:127-136, 193-204. For real
    element types every conj is the identity and nothing is launched or allocated."""
    if res.dtype.is_complex:
        conj_into(res, res)
    vc = conj_into(torch.empty_like(v), v) if v.dtype.is_complex else v   # conj.(v) allocates in the reference too
    _call_t(res, f, vc, conj_scalar(alpha), conj_scalar(beta), p)
    if res.dtype.is_complex:
        conj_into(res, res)
    return touched(res)


def _mul_adjoint(res, op, v, alpha, beta):
    """src/adjtrans.jl:90-137."""
    p = op.parent
    if not (v.shape[0] == p.size(1) and res.shape[0] == p.size(2)):
        raise LinearOperatorException("shape mismatch")
    if ishermitian(p):
        return mul(res, p, v, alpha, beta)
    if p.ctprod is not None:
        p.nctprod += _COUNT
        return _call_t(res, p.ctprod, v, alpha, beta, p)
    tprod = p.tprod
    if tprod is None:
        if issymmetric(p):
            p.nprod += _COUNT
            tprod = p.prod
        else:
            raise LinearOperatorException("unable to infer conjugate transpose operator")
    else:
        p.ntprod += _COUNT
    return _call_conj_sandwich(res, tprod, v, alpha, beta, p)


def _mul_transpose(res, op, v, alpha, beta):
    """src/adjtrans.jl:158-205."""
    p = op.parent
    if not (v.shape[0] == p.size(1) and res.shape[0] == p.size(2)):
        raise LinearOperatorException("shape mismatch")
    if issymmetric(p):
        return mul(res, p, v, alpha, beta)
    if p.tprod is not None:
        p.ntprod += _COUNT
        return _call_t(res, p.tprod, v, alpha, beta, p)
    ctprod = p.ctprod
    if ctprod is None:
        if ishermitian(p):
            p.nprod += _COUNT
            ctprod = p.prod
        else:
            raise LinearOperatorException("unable to infer transpose operator")
    else:
        p.nctprod += _COUNT
    return _call_conj_sandwich(res, ctprod, v, alpha, beta, p)


def apply(op, v: torch.Tensor) -> torch.Tensor:
    """`op * v` — src/operations.jl:43-48: res = similar(v, promote_type(T,S), nrow); mul!(res, op, v)."""
    check_vec(v, "v")
    T = op.eltype if (op.eltype.is_floating_point or op.eltype.is_complex) else v.dtype
    res = torch.empty(op.size(1), dtype=torch.promote_types(T, v.dtype), device=v.device)
    mul(res, op, v)
    return res


# ----------------------------------------------------------------------------- lazy wrappers
class _Wrapper(AbstractLinearOperator):
    def __init__(self, parent):
        self.parent = parent
        self.eltype = parent.eltype


class AdjointLinearOperator(_Wrapper):
    nrow = property(lambda s: s.parent.size(2))
    ncol = property(lambda s: s.parent.size(1))
    symmetric = property(lambda s: issymmetric(s.parent))
    hermitian = property(lambda s: ishermitian(s.parent))


class TransposeLinearOperator(_Wrapper):
    nrow = property(lambda s: s.parent.size(2))
    ncol = property(lambda s: s.parent.size(1))
    symmetric = property(lambda s: issymmetric(s.parent))
    hermitian = property(lambda s: ishermitian(s.parent))


class ConjugateLinearOperator(_Wrapper):
    nrow = property(lambda s: s.parent.size(1))
    ncol = property(lambda s: s.parent.size(2))
    symmetric = property(lambda s: issymmetric(s.parent))
    hermitian = property(lambda s: ishermitian(s.parent))


class UniversalEye:
    """`opEye()` — src/special-operators.jl:5-34: the size-less identity. `op * x` IS `x` (vector, matrix or operator: the
    very same object), `x * op` likewise; adjoint / transpose / conj return the operator itself. Pure host logic."""
    _inst = None

    def __new__(cls):
        if cls._inst is None:
            cls._inst = super().__new__(cls)
        return cls._inst                                   # opEye() === opEye()

    __array_priority__ = 1000                              # `ndarray * op` defers to __rmul__

    def __mul__(self, other):
        return other

    __rmul__ = __mul__
    __matmul__ = __mul__
    __rmatmul__ = __mul__
    T = property(lambda s: s)
    H = property(lambda s: s)

    def __repr__(self):
        return "Identity operator"


def adjoint(A):      # src/adjtrans.jl:33-45
    if isinstance(A, UniversalEye):
        return A                                           # special-operators.jl:27-29
    if isinstance(A, AdjointLinearOperator):
        return A.parent
    if isinstance(A, ConjugateLinearOperator):
        return transpose(A.parent)
    if isinstance(A, TransposeLinearOperator):
        return conj(A.parent)
    return AdjointLinearOperator(A)


def transpose(A):
    if isinstance(A, UniversalEye):
        return A                                           # special-operators.jl:27-29
    if isinstance(A, TransposeLinearOperator):
        return A.parent
    if isinstance(A, AdjointLinearOperator):
        return conj(A.parent)
    if isinstance(A, ConjugateLinearOperator):
        return adjoint(A.parent)
    return TransposeLinearOperator(A)


def conj(A):
    if isinstance(A, UniversalEye):
        return A                                           # special-operators.jl:27-29
    if isinstance(A, ConjugateLinearOperator):
        return A.parent
    if isinstance(A, AdjointLinearOperator):
        return transpose(A.parent)
    if isinstance(A, TransposeLinearOperator):
        return adjoint(A.parent)
    return ConjugateLinearOperator(A)


# ----------------------------------------------------------------------------- combinators
def _promote_eltype(*ops):
    dt = None
    for o in ops:
        t = o.eltype if not isinstance(o, torch.dtype) else o
        dt = t if dt is None else torch.promote_types(dt, t)
    return dt


def neg(op):
    """-op — src/operations.jl:102-115 (wrappers: src/adjtrans.jl:263-265)."""
    if isinstance(op, AdjointLinearOperator):
        return adjoint(neg(op.parent))
    if isinstance(op, TransposeLinearOperator):
        return transpose(neg(op.parent))
    if isinstance(op, ConjugateLinearOperator):
        return conj(neg(op.parent))
    prod = matrixwise(lambda res, v, a, b: mul(res, op, v, -a, b))
    tprod = matrixwise(lambda res, u, a, b: mul(res, transpose(op), u, -a, b))
    ctprod = matrixwise(lambda res, w, a, b: mul(res, adjoint(op), w, -a, b))
    out = LinearOperator(op.eltype, op.nrow, op.ncol, op.symmetric, op.hermitian, prod, tprod, ctprod,
                         S=storage_type(op))
    out._deps = (op,)
    return out


def prod_op(res, op1, op2, vtmp, v, alpha, beta):
    """prod_op! — src/operations.jl:117-128."""
    mul(vtmp, op2, v)
    mul(res, op1, vtmp, alpha, beta)


def compose(op1, op2):
    """op1 * op2 — src/operations.jl:131-156."""
    T = _promote_eltype(op1, op2)
    m1, n1 = op1.shape
    m2, n2 = op2.shape
    if m2 != n1:
        raise LinearOperatorException("shape mismatch")
    S = promote_storage(storage_type(op1), storage_type(op2))
    if not (S.dtype.is_floating_point or S.dtype.is_complex):      # index-typed restriction composed with restriction
        S = Storage(torch.float64, S.device)
    vtmp, utmp, wtmp = S.zeros(m2), S.zeros(n1), S.zeros(n1)
    prod = lambda res, v, a, b: prod_op(res, op1, op2, vtmp, v, a, b)
    tprod = lambda res, u, a, b: prod_op(res, transpose(op2), transpose(op1), utmp, u, a, b)
    ctprod = lambda res, w, a, b: prod_op(res, adjoint(op2), adjoint(op1), wtmp, w, a, b)
    out = LinearOperator(T, m1, n2, False, False, prod, tprod, ctprod, S=S)
    out._deps = (op1, op2)
    return out


def scale_op(op, x):
    """op * x, x * op — src/operations.jl:163-183; wrappers src/adjtrans.jl:267-273."""
    if isinstance(op, AdjointLinearOperator):
        return adjoint(scale_op(op.parent, conj_scalar(x)))         # adjoint(op.parent * conj(x)) (src/adjtrans.jl:266)
    if isinstance(op, TransposeLinearOperator):
        return transpose(scale_op(op.parent, x))
    if isinstance(op, ConjugateLinearOperator):
        return conj(scale_op(op.parent, conj_scalar(x)))            # conj(op.parent * conj(x)) (src/adjtrans.jl:268)
    T = op.eltype
    prod = matrixwise(lambda res, v, a, b: mul(res, op, v, x * a, b))
    tprod = matrixwise(lambda res, u, a, b: mul(res, transpose(op), u, x * a, b))
    xc = conj_scalar(x)
    ctprod = matrixwise(lambda res, w, a, b: mul(res, adjoint(op), w, xc * a, b))  # x' (src/operations.jl:166)
    isreal_x = (not _is_complex_scalar(x)) or complex(x).imag == 0      # isreal(x) (src/operations.jl:172)
    if _is_complex_scalar(x):                                            # T = promote_type(eltype(op), typeof(x))
        T = torch.promote_types(T, torch.complex128 if _is_f64_scalar(x) else torch.complex64)
    out = LinearOperator(T, op.nrow, op.ncol, op.symmetric, op.hermitian and isreal_x, prod, tprod, ctprod,
                         S=storage_type(op))
    out._deps = (op,)
    return out


def sum_prod(res, op1, op2, v, alpha, beta):
    """sum_prod! — src/operations.jl:187-197."""
    mul(res, op1, v, alpha, beta)
    mul(res, op2, v, alpha, one(op2.eltype))


def add(op1, op2):
    """op1 + op2 — src/operations.jl:199-215."""
    m1, n1 = op1.shape
    m2, n2 = op2.shape
    if m1 != m2 or n1 != n2:
        raise LinearOperatorException("shape mismatch")
    T = _promote_eltype(op1, op2)
    prod = matrixwise(lambda res, v, a, b: sum_prod(res, op1, op2, v, a, b))
    tprod = matrixwise(lambda res, u, a, b: sum_prod(res, transpose(op1), transpose(op2), u, a, b))
    ctprod = matrixwise(lambda res, w, a, b: sum_prod(res, adjoint(op1), adjoint(op2), w, a, b))
    symm = issymmetric(op1) and issymmetric(op2)
    herm = ishermitian(op1) and ishermitian(op2)
    S = promote_storage(storage_type(op1), storage_type(op2))
    out = LinearOperator(T, m1, n1, symm, herm, prod, tprod, ctprod, S=S)
    out._deps = (op1, op2)
    return out


# ----------------------------------------------------------------------------- cat (src/cat.jl)
def hcat_prod(res, A, B, Ancol, nV, v, alpha, beta):
    """hcat_prod! — src/cat.jl:7-19 (views are pointer+offset: torch slices)."""
    mul(res, A, v[:Ancol], alpha, beta)
    mul(res, B, v[Ancol:nV], alpha, one(B.eltype))


def hcat_ctprod(res, A, B, Ancol, nV, u, alpha, beta):
    """hcat_ctprod! — src/cat.jl:21-33."""
    mul(res[:Ancol], A, u, alpha, beta)
    mul(res[Ancol:nV], B, u, alpha, beta)


def _hcat2(A, B):
    """hcat(A, B) — src/cat.jl:35-51."""
    if A.size(1) != B.size(1):
        raise LinearOperatorException("hcat: inconsistent row sizes")
    nrow = A.size(1)
    Ancol, Bncol = A.size(2), B.size(2)
    T = _promote_eltype(A, B)
    prod = lambda res, v, a, b: hcat_prod(res, A, B, Ancol, Ancol + Bncol, v, a, b)
    tprod = lambda res, u, a, b: hcat_ctprod(res, transpose(A), transpose(B), Ancol, Ancol + Bncol, u, a, b)
    ctprod = lambda res, w, a, b: hcat_ctprod(res, adjoint(A), adjoint(B), Ancol, Ancol + Bncol, w, a, b)
    S = promote_storage(storage_type(A), storage_type(B))
    out = LinearOperator(T, nrow, Ancol + Bncol, False, False, prod, tprod, ctprod, S=S)
    out._deps = (A, B)
    return out


def _as_op(x):
    if isinstance(x, AbstractLinearOperator):
        return x
    if (isinstance(x, torch.Tensor) and x.dim() == 2) or hasattr(x, "tocsc"):
        from .leaves import LinearOperatorFromMatrix
        return LinearOperatorFromMatrix(x)             # dense or sparse (torch.sparse_csc / _csr / _coo, scipy.sparse)
    raise TypeError(f"cannot concatenate {type(x)}")


def hcat(*ops):
    """hcat(ops...) folds left — src/cat.jl:53-59."""
    ops = [_as_op(o) for o in ops]
    op = ops[0]
    for o in ops[1:]:
        op = _hcat2(op, o)
    return op


def vcat_prod(res, A, B, Anrow, nV, u, alpha, beta):
    """vcat_prod! — src/cat.jl:65-77."""
    mul(res[:Anrow], A, u, alpha, beta)
    mul(res[Anrow:nV], B, u, alpha, beta)


def vcat_ctprod(res, A, B, Anrow, nV, v, alpha, beta):
    """vcat_ctprod! — src/cat.jl:79-91."""
    mul(res, A, v[:Anrow], alpha, beta)
    mul(res, B, v[Anrow:nV], alpha, one(B.eltype))


def _vcat2(A, B):
    """vcat(A, B) — src/cat.jl:93-109."""
    if A.size(2) != B.size(2):
        raise LinearOperatorException("vcat: inconsistent column sizes")
    Anrow, Bnrow = A.size(1), B.size(1)
    ncol = A.size(2)
    T = _promote_eltype(A, B)
    prod = lambda res, v, a, b: vcat_prod(res, A, B, Anrow, Anrow + Bnrow, v, a, b)
    tprod = lambda res, u, a, b: vcat_ctprod(res, transpose(A), transpose(B), Anrow, Anrow + Bnrow, u, a, b)
    ctprod = lambda res, w, a, b: vcat_ctprod(res, adjoint(A), adjoint(B), Anrow, Anrow + Bnrow, w, a, b)
    S = promote_storage(storage_type(A), storage_type(B))
    out = LinearOperator(T, Anrow + Bnrow, ncol, False, False, prod, tprod, ctprod, S=S)
    out._deps = (A, B)
    return out


def vcat(*ops):
    """vcat(ops...) folds left — src/cat.jl:111-117."""
    ops = [_as_op(o) for o in ops]
    op = ops[0]
    for o in ops[1:]:
        op = _vcat2(op, o)
    return op


def hvcat(rows, *ops):
    """hvcat(rows, ops...) — src/cat.jl:120-129: vcat of hcats."""
    rs, a = [], 0
    for r in rows:
        rs.append(hcat(*ops[a:a + r]))
        a += r
    return vcat(*rs)


# ----------------------------------------------------------------------------- Matrix(op)
def to_dense(op) -> torch.Tensor:
    """`Matrix(op)` — src/abstract.jl:282-292: ncol products with unit vectors."""
    m, n = op.shape
    S = storage_type(op)
    dt = op.eltype if (op.eltype.is_floating_point or op.eltype.is_complex) else torch.float64
    A = torch.empty((n, m), dtype=dt, device=S.device)   # row i = column i of the operator
    ei = torch.zeros(n, dtype=dt, device=S.device)
    for i in range(n):
        ei[i] = 1
        mul(A[i], op, ei)
        ei[i] = 0
    return A.t()          # an m x n COLUMN-MAJOR view (Julia's Matrix layout): no second copy
