"""Quasi-Newton operators: `LBFGSOperator`, `InverseLBFGSOperator`, `LSR1Operator`, `push!`,
`diag`, `reset!`, `solve_shifted_system!`, `ldiv!` — the host side of `mxlo_qn_*`.

Mirrors src/lbfgs.jl, src/lsr1.jl and src/utilities.jl:207-289. The operator objects satisfy the
reference's *structural* contract for quasi-Newton operators (fields nrow ncol symmetric hermitian
prod tprod ctprod nprod ntprod nctprod + `data`, src/lbfgs.jl:62-75, src/lsr1.jl:39-51), so `mul`,
counters, `size`, flags and the combinators of :mod:`operators` work on them unchanged. The state
(`s`, `y`, `a`, `b` panels, Gram matrices, scalars) lives in HBM inside the C handle.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib
from .device import Storage, check_vec, dtype_code, get_ctx, indexed_device, ptr
from .operators import AbstractLinearOperator, LinearOperatorException, scalar_flags, touched


class _QNData:
    """`LBFGSData` / `LSR1Data` view (src/lbfgs.jl:4-24, src/lsr1.jl:4-17): the scalar fields the
    reference's tests read, fetched from the device handle on access."""

    def __init__(self, op):
        self._op = op

    def _scalars(self):
        op = self._op
        sc = (C.c_double * 5)()
        ys = (C.c_double * op.mem)()
        aux = (C.c_double * op.mem)()
        _lib.call("mxlo_qn_get_scalars", op._h, sc, ys, aux)
        return list(sc), np.array(ys[:]), np.array(aux[:])

    @property
    def insert(self) -> int:            # 1-based like Julia
        return int(self._scalars()[0][0])

    @property
    def scaling_factor(self) -> float:
        return self._scalars()[0][1]

    @property
    def opnorm_upper_bound(self) -> float:
        return self._scalars()[0][2]

    @property
    def mem(self) -> int:
        return self._op.mem

    @property
    def scaling(self) -> bool:
        return self._op.scaling

    @property
    def damped(self) -> bool:
        return self._op.damped

    @property
    def ys(self) -> np.ndarray:
        return self._scalars()[1]

    def column(self, which: str, k: int) -> torch.Tensor:
        """Copy of panel column `k` (0-based slot) of 's' | 'y' | 'a' | 'b' (test / debug aid)."""
        op = self._op
        p = C.c_void_p()
        _lib.call("mxlo_qn_column", op._h, {"s": 0, "y": 1, "a": 2, "b": 3}[which], k, C.byref(p))
        out = torch.empty(op.nrow, dtype=op.eltype, device=op.S.device)
        _lib.call("mxlo_memcpy_d2d", op._ctx.handle, ptr(out), p, out.numel() * out.element_size())
        return out


class _QNOperator(AbstractLinearOperator):
    _has_args5 = True       # has_args5(op) = isallocated5(op) = true (src/lbfgs.jl:102-103)

    def __init__(self, kind: int, T: torch.dtype, n: int, mem: int, scaling: bool, damped: bool, sigma2: float,
                 sigma3: float, device):
        dev = indexed_device(device)
        self._ctx = get_ctx(dev)
        self._kind = kind
        self.eltype = T
        self.nrow = self.ncol = int(n)
        self.symmetric = self.hermitian = True
        self.mem = max(int(mem), 1)
        self.scaling, self.damped = bool(scaling), bool(damped)
        self.S = Storage(T, dev)        # storage_type(op) (the reference hard-codes Vector{T}, lbfgs.jl:104)
        self.nprod = self.ntprod = self.nctprod = 0
        self._h = C.c_void_p()
        _lib.call("mxlo_qn_create", self._ctx.handle, kind, dtype_code(T), int(n), self.mem, int(scaling), int(damped),
                  float(sigma2), float(sigma3), C.byref(self._h))
        self.data = _QNData(self)
        self._pending_shift = None       # set by ShiftedOperator around one mul! (fused axpy!, mxlo_qn_mul_shifted)
        self._nupdate = 0                # bumped by every accepted push! and every reset! (operators.state_version)
        self.inverse = kind == _lib.QN_LBFGS_INV
        prod = lambda res, x, a, b: self._multiply(res, x, a, b)
        self.prod = prod
        if kind == _lib.QN_LSR1:        # tprod! = ctprod! = nothing with symmetric = hermitian = true (lsr1.jl:110)
            self.tprod = self.ctprod = None
        else:                            # prod! three times (lbfgs.jl:157,205)
            self.tprod = self.ctprod = prod

    def _multiply(self, res, x, alpha, beta):
        """lbfgs_multiply / lsr1_multiply (src/lbfgs.jl:117-154,173-202; src/lsr1.jl:89-107)."""
        check_vec(res, "res", self.eltype)
        check_vec(x, "x", self.eltype)
        self._ctx.bind_stream()
        if self._pending_shift is not None:
            _lib.call("mxlo_qn_mul_shifted", self._h, ptr(res), ptr(x), float(alpha), float(beta),
                      self._pending_shift, scalar_flags(self.eltype, alpha, beta))
            return
        _lib.call("mxlo_qn_mul", self._h, ptr(res), ptr(x), float(alpha), float(beta),
                  scalar_flags(self.eltype, alpha, beta))

    def set_mode(self, mode: str):
        """'twopass' (panel form, default) or 'reforder' (reference statement order) for the inverse two-loop."""
        _lib.call("mxlo_qn_set_mode", self._h, {"twopass": _lib.INV_TWOPASS, "reforder": _lib.INV_REFORDER}[mode])
        return self

    def set_push_mode(self, mode: str):
        """'gram' (Gram-matrix recurrence + one panel pass that forms the a_k), 'reforder' (the reference's O(m²)
        statement order) or 'compact' (forward L-BFGS: Gram recurrence, a_k left implicit — push! costs only its
        3m dots; diag!/solve_shifted_system! materialise the panel on demand)."""
        _lib.call("mxlo_qn_set_push_mode", self._h,
                  {"gram": _lib.PUSH_GRAM, "reforder": _lib.PUSH_REFORDER, "compact": _lib.PUSH_COMPACT}[mode])
        return self

    def _reset_data(self):
        _lib.call("mxlo_qn_reset", self._h)
        self._nupdate += 1

    def _state_version(self):
        return ("qn", id(self), self._nupdate)

    def _check_len(self, **vecs):
        """DimensionMismatch of the reference's broadcasts / BLAS calls: the C side writes exactly n elements."""
        for name, t in vecs.items():
            if t.numel() != self.nrow:
                raise LinearOperatorException(f"shape mismatch: {name} has {t.numel()} elements, the operator is "
                                              f"{self.nrow} x {self.ncol}")

    def __del__(self):
        try:
            if self._h:
                _lib.lib().mxlo_qn_destroy(self._h)
        except Exception:
            pass


class LBFGSOperatorType(_QNOperator):
    pass


class LSR1OperatorType(_QNOperator):
    pass


def _targs(T, n):
    if isinstance(T, int):          # LBFGSOperator(n; kwargs...) -> Float64
        return torch.float64, T
    return T, n


def _kw(kw, sigma2, sigma3):
    """The reference's keyword spellings (σ₂, σ₃; `inverse` is fixed by the constructor name); anything else
    is an error, as a Julia keyword typo would be (MethodError)."""
    kw.pop("inverse", None)
    s2, s3 = kw.pop("σ₂", sigma2), kw.pop("σ₃", sigma3)
    if kw:
        raise TypeError(f"unsupported keyword argument(s): {sorted(kw)}")
    return s2, s3


def InverseLBFGSOperator(T, n: Optional[int] = None, mem: int = 5, scaling: bool = True, damped: bool = False,
                         sigma2: float = 0.99, sigma3: float = 10.0, device=None, **kw):
    """InverseLBFGSOperator(T, n; mem=5, scaling=true, damped=false, σ₂=0.99, σ₃=10.0) — src/lbfgs.jl:106-160."""
    T, n = _targs(T, n)
    s2, s3 = _kw(kw, sigma2, sigma3)
    return LBFGSOperatorType(_lib.QN_LBFGS_INV, T, n, mem, scaling, damped, s2, s3, device)


def LBFGSOperator(T, n: Optional[int] = None, mem: int = 5, scaling: bool = True, damped: bool = False,
                  sigma2: float = 0.99, sigma3: float = 10.0, device=None, **kw):
    """LBFGSOperator(T, n; mem=5, scaling=true, ...) forward form — src/lbfgs.jl:162-208."""
    T, n = _targs(T, n)
    s2, s3 = _kw(kw, sigma2, sigma3)
    return LBFGSOperatorType(_lib.QN_LBFGS_FWD, T, n, mem, scaling, damped, s2, s3, device)


def LSR1Operator(T, n: Optional[int] = None, mem: int = 5, scaling: bool = True, device=None):
    """LSR1Operator(T, n; mem=5, scaling=true) — src/lsr1.jl:80-113 (LSR1Data default scaling=true, :19)."""
    T, n = _targs(T, n)
    return LSR1OperatorType(_lib.QN_LSR1, T, n, mem, scaling, False, 0.99, 10.0, device)


def push(op, s: torch.Tensor, y: torch.Tensor, *args):
    """push!(op, s, y) | push!(op, s, y, Bs) | push!(op, s, y, α, g) | push!(op, s, y, α, g, Bs)
    — src/lbfgs.jl:257-367, src/lsr1.jl:115-184. Returns `op` like the reference."""
    if hasattr(op, "_push"):                              # diagonal quasi-Newton family (diagqn.py)
        if args:
            raise TypeError("push!(::AbstractDiagonalQuasiNewtonOperator, s, y) takes no extra arguments")
        return op._push(s, y)
    if not isinstance(op, _QNOperator):
        raise TypeError("push! is defined for quasi-Newton operators")
    check_vec(s, "s", op.eltype)
    check_vec(y, "y", op.eltype)
    if s.numel() != op.nrow or y.numel() != op.nrow:
        raise LinearOperatorException("shape mismatch")
    op._ctx.bind_stream()
    acc = C.c_int32(0)
    if isinstance(op, LSR1OperatorType):
        if args:
            raise TypeError("push!(::LSR1Operator, s, y) takes no extra arguments")
        _lib.call("mxlo_qn_push", op._h, ptr(s), ptr(y), C.byref(acc))
        op._last_push_accepted = bool(acc.value)
        op._nupdate += int(acc.value)
        return op
    if len(args) == 0:
        if op.damped:                                   # :273-275 — push!(op, s, y, similar(s))
            if op.inverse:
                raise RuntimeError("This function be used for forward operators. Use push!(op, s, y, α, g, Bs) instead.")
            return push(op, s, y, torch.empty_like(s))
        _lib.call("mxlo_qn_push", op._h, ptr(s), ptr(y), C.byref(acc))
    elif len(args) == 1:                                # push!(op, s, y, Bs) :289-323
        Bs, = args
        if not op.damped:
            raise RuntimeError("This push! should be used for damped operators")
        if op.inverse:
            raise RuntimeError("This function be used for forward operators. Use push!(op, s, y, α, g, Bs) instead.")
        op._check_len(Bs=check_vec(Bs, "Bs", op.eltype))
        _lib.call("mxlo_qn_push_damped_fwd", op._h, ptr(s), ptr(y), ptr(Bs), C.byref(acc))
        touched(Bs)
    elif len(args) in (2, 3):                           # push!(op, s, y, α, g[, Bs]) :325-367
        alpha, g = args[0], args[1]
        Bs = args[2] if len(args) == 3 else torch.empty_like(g)
        if not op.damped:
            raise RuntimeError("This push! should be used for damped operators")
        if not op.inverse:
            raise RuntimeError("This function be used for inverse operators. Use push!(op, s, y, Bs) instead.")
        op._check_len(g=check_vec(g, "g", op.eltype), Bs=check_vec(Bs, "Bs", op.eltype))
        _lib.call("mxlo_qn_push_damped_inv", op._h, ptr(s), ptr(y), float(alpha), ptr(g), ptr(Bs), C.byref(acc))
        touched(Bs)
        touched(y)                                      # Powell damping overwrites the caller's y (:351)
    else:
        raise TypeError("push!: wrong number of arguments")
    op._last_push_accepted = bool(acc.value)
    op._nupdate += int(acc.value)
    return op


def diag(op, d: torch.Tensor | None = None) -> torch.Tensor:
    """diag(op) / diag!(op, d) — src/lbfgs.jl:369-395 (forward only), src/lsr1.jl:186-211. With `d` the call is the
    in-place `diag!` (nothing is allocated); without, `diag(op)` allocates the result like the reference (:371)."""
    if isinstance(op, LBFGSOperatorType) and op.inverse:
        raise LinearOperatorException("only the diagonal of a forward L-BFGS approximation is available")
    if d is None:
        d = torch.empty(op.nrow, dtype=op.eltype, device=op.S.device)
    else:
        op._check_len(d=check_vec(d, "d", op.eltype))
    op._ctx.bind_stream()
    _lib.call("mxlo_qn_diag", op._h, ptr(d))
    return touched(d)


def solve_shifted_system(x: torch.Tensor, B, b: torch.Tensor, sigma: float) -> torch.Tensor:
    """solve_shifted_system!(x, B, b, σ) — src/utilities.jl:207-248. Returns `x` (the same object)."""
    if not (isinstance(B, LBFGSOperatorType) and not B.inverse):
        raise TypeError("solve_shifted_system! is defined for forward LBFGSOperator")
    if sigma < 0:
        raise ValueError("σ must be nonnegative")       # ArgumentError, :213-215
    B._check_len(x=check_vec(x, "x", B.eltype), b=check_vec(b, "b", B.eltype))
    B._ctx.bind_stream()
    _lib.call("mxlo_qn_solve_shifted", B._h, ptr(x), ptr(b), float(sigma))
    return touched(x)


def ldiv(x: torch.Tensor, B, b: torch.Tensor) -> torch.Tensor:
    """ldiv!(x, B, b) — src/utilities.jl:281-289."""
    return solve_shifted_system(x, B, b, 0.0)
