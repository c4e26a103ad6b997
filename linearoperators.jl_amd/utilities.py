"""Host-side diagnostics of the reference's src/utilities.jl:1-149 — `normest`, `check_ctranspose`, `check_hermitian`,
`check_positive_definite` — as CALLERS of the `mul!` hot path: every product `op * v` / `op' * v` runs in libmxlo.so, the
inner products between the resulting device vectors run there too (`mxlo_dot`, `mxlo_dot_c`: LinearAlgebra.dot, conj on
the first argument), and only the handful of scalars they produce come back to the host. Same tests, thresholds and return
values as the reference; Integer operators (`utilities.jl:75-82, 104-114, 137-144`) have no device storage type here and
raise. The random probe vectors come from torch's generator of the operator's device instead of Julia's `rand`."""
from __future__ import annotations

import math

import torch

from . import _lib
from .device import ctx_of, dtype_code, ptr
from .operators import AbstractLinearOperator, LinearOperatorException, adjoint, apply, storage_type

__all__ = ["normest", "check_ctranspose", "check_hermitian", "check_positive_definite"]


def _as_operator(S):
    if isinstance(S, AbstractLinearOperator):
        return S
    from .leaves import LinearOperatorFromMatrix              # check_*(M::AbstractMatrix) = check_*(LinearOperator(M))
    return LinearOperatorFromMatrix(S)


def _float_op(op):
    T = op.eltype
    if not (T.is_floating_point or T.is_complex):
        raise TypeError("Integer operators have no device storage type (src/utilities.jl:75-82 is host-only)")
    return T


def _eps(T: torch.dtype) -> float:
    return torch.finfo(T).eps                                 # eps(real(eltype(op))): finfo of a complex dtype is its real part's


def _dot(a: torch.Tensor, b: torch.Tensor):
    """LinearAlgebra.dot(a, b) = Σ conj(a_i) b_i of two device vectors, reduced in libmxlo.so; a Python float / complex."""
    if a.dtype != b.dtype or a.numel() != b.numel():
        raise LinearOperatorException("shape mismatch")
    ctx = ctx_of(a)
    ctx.bind_stream()
    out = torch.empty(2, dtype=torch.float64, device=a.device)
    if a.dtype.is_complex:
        _lib.call("mxlo_dot_c", ctx.handle, dtype_code(a.dtype, complex_ok=True), ptr(a), ptr(b), a.numel(), ptr(out))
        re, im = out.tolist()
        return complex(re, im)
    _lib.call("mxlo_dot", ctx.handle, dtype_code(a.dtype), ptr(a), ptr(b), a.numel(), ptr(out))
    return float(out[0])


def _norm(a: torch.Tensor) -> float:
    d = _dot(a, a)
    return math.sqrt(d.real if isinstance(d, complex) else d)


def _rand(op, n: int, gen=None) -> torch.Tensor:
    """rand(n) (uniform on [0, 1), real) in the operator's storage type"""
    S = storage_type(op)
    T = _float_op(op)
    real = torch.rand(n, dtype=torch.float64, device=S.device, generator=gen)
    return real.to(T)


def normest(S, tol: float = -1, maxiter: int = 100, generator=None):
    """normest(S, tol = -1, maxiter = 100) -> (e, cnt): estimate of the matrix 2-norm by the power iteration on S'S of
    Matlab's NORMEST — src/utilities.jl:20-58. Two applies per iteration, both on the device."""
    op = _as_operator(S)
    T = _float_op(op)
    m, n = op.size()
    dev = storage_type(op).device
    cnt = 0
    if tol == -1:
        tol = float(torch.finfo(torch.float64).eps if T in (torch.float64, torch.complex128) else torch.finfo(T).eps)  # Float64(eps(eltype(S)))
    v = torch.ones(m, dtype=T, device=dev)                                                  # :27-28
    v[torch.randn(m, device=dev, generator=generator) < 0] = -1
    x = apply(adjoint(op), v)                                                               # mul!(x, S', v)  :30
    e = _norm(x)
    if e == 0:                                                                              # :33-35
        return e, cnt
    x.div_(e)
    e_0 = 0.0
    while abs(e - e_0) > tol * e:                                                           # :41
        e_0 = e
        Sx = apply(op, x)                                                                   # :43
        if int(torch.count_nonzero(Sx)) == 0:                                               # :44-46
            Sx = torch.randn(m, dtype=torch.float64, device=dev, generator=generator).to(T)
        x = apply(adjoint(op), Sx)                                                          # :47
        normx = _norm(x)
        e = normx / _norm(Sx)                                                               # :49
        x.div_(normx)
        cnt += 1
        if cnt > maxiter:                                                                   # :52-55
            import warnings
            warnings.warn(f"normest did not converge (maxiter = {maxiter}, tol = {tol})")
            break
    return e, cnt


def check_ctranspose(op, generator=None) -> bool:
    """Cheap check that the operator and its conjugate transpose are related: |y'(Ax) − conj(x'(A'y))| small —
    src/utilities.jl:65-73."""
    op = _as_operator(op)
    T = _float_op(op)
    m, n = op.size()
    x, y = _rand(op, n, generator), _rand(op, m, generator)
    yAx = _dot(y, apply(op, x))
    xAty = _dot(x, apply(adjoint(op), y))
    eps = _eps(T)
    return abs(yAx - (xAty.conjugate() if isinstance(xAty, complex) else xAty)) < (abs(yAx) + eps) * eps ** (1 / 3)


def check_hermitian(op, generator=None) -> bool:
    """Cheap check that the operator is Hermitian: (Av)'(Av) against v'(A(Av)) — src/utilities.jl:91-102."""
    op = _as_operator(op)
    T = _float_op(op)
    m, n = op.size()
    if m != n:
        raise LinearOperatorException("shape mismatch")
    v = _rand(op, n, generator)
    w = apply(op, v).clone()                      # copy necessary to guard against in-place operators (:96)
    s = _dot(w, w)
    y = apply(op, w)
    t = _dot(v, y)
    eps = _eps(T)
    return abs(s - t) < (abs(s) + eps) * eps ** (1 / 3)


def check_positive_definite(op, semi: bool = False, generator=None) -> bool:
    """Cheap check that the operator is positive (semi-)definite: v'(Av) for one random v — src/utilities.jl:123-135."""
    op = _as_operator(op)
    T = _float_op(op)
    m, n = op.size()
    if m != n:
        raise LinearOperatorException("shape mismatch")
    v = _rand(op, n, generator)
    vw = _dot(v, apply(op, v))
    eps = _eps(T)
    if isinstance(vw, complex):
        if vw.imag > math.sqrt(eps) * abs(vw):
            return False
        vw = vw.real
    return vw >= 0 if semi else vw > 0
