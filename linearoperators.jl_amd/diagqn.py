"""Diagonal quasi-Newton operators — src/DiagonalHessianApproximation.jl.

`DiagonalPSB(d)`, `DiagonalAndrei(d)`, `DiagonalBFGS(d)`, `SpectralGradient(σ, n)`: the `mul!` is
`mulSquareOpDiagonal!` on the operator's own `d` (:37,112,179,226; SpectralGradient keeps a 1-element `d`),
`push!(B, s, y)` is ONE fused reduction pass + the reference's scalar recurrence + ONE update pass in
libmxlo.so (`mxlo_diagqn_push`), `reset!` sets `d .= 1` (:71-77). Everything that touches a vector runs on
the GPU; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .device import Storage, check_vec, dtype_code, get_ctx, indexed_device, ptr
from .leaves import mulSquareOpDiagonal
from .operators import touched, AbstractLinearOperator


class _DiagonalQN(AbstractLinearOperator):
    """AbstractDiagonalQuasiNewtonOperator: fields d nrow ncol symmetric hermitian prod! tprod! ctprod! + counters."""
    _has_args5 = True                                   # isallocated5 = has_args5 = true (:250-255)
    _kind = None

    def __init__(self, d: torch.Tensor, n: int):
        self.d = d
        self.eltype = d.dtype
        self.nrow = self.ncol = int(n)
        self.symmetric = self.hermitian = True
        self.S = Storage(d.dtype, d.device)
        self.nprod = self.ntprod = self.nctprod = 0
        prod = lambda res, v, a, b: mulSquareOpDiagonal(res, self.d, v, a, b)
        self.prod = self.tprod = self.ctprod = prod

    def _push(self, s: torch.Tensor, y: torch.Tensor):
        check_vec(s, "s", self.eltype)
        check_vec(y, "y", self.eltype)
        if s.numel() != self.nrow or y.numel() != self.nrow:
            raise ValueError("push!: s and y must have the operator's size")
        ctx = get_ctx(self.d.device)
        st = C.c_int32(0)
        _lib.call("mxlo_diagqn_push", ctx.handle, dtype_code(self.eltype), self._kind, ptr(self.d), ptr(s), ptr(y),
                  self.nrow, C.byref(st))
        if st.value != 0:
            raise RuntimeError(self._zero_msg)           # ErrorException in the reference
        touched(self.d)                                  # d changed behind torch's back: bump its version counter
        return self

    _zero_msg = "Cannot update DiagonalQN operator with s=0"

    def _reset_data(self):                               # op.d .= one(T) (:72)
        ctx = get_ctx(self.d.device)
        _lib.call("mxlo_fill", ctx.handle, dtype_code(self.eltype), ptr(self.d), self.d.numel(), 1.0)
        touched(self.d)

    @property
    def _deps(self):
        return (self.d,)


class DiagonalPSBType(_DiagonalQN):
    _kind = _lib.DQN_PSB


class DiagonalAndreiType(_DiagonalQN):
    _kind = _lib.DQN_ANDREI


class DiagonalBFGSType(_DiagonalQN):
    _kind = _lib.DQN_BFGS


class SpectralGradientType(_DiagonalQN):
    _kind = _lib.DQN_SPECTRAL
    _zero_msg = "Cannot divide by zero and s .= 0"


def _own(d: torch.Tensor) -> torch.Tensor:
    check_vec(d, "d")
    dtype_code(d.dtype)
    return d


def DiagonalPSB(d: torch.Tensor):
    """DiagonalPSB(d) (:24-43): `d` is the operator's diagonal itself (not copied), updated in place by push!."""
    return DiagonalPSBType(_own(d), d.numel())


def DiagonalAndrei(d: torch.Tensor):
    """DiagonalAndrei(d) (:96-115)."""
    return DiagonalAndreiType(_own(d), d.numel())


def DiagonalBFGS(d: torch.Tensor):
    """DiagonalBFGS(d) (:210-228)."""
    return DiagonalBFGSType(_own(d), d.numel())


def SpectralGradient(sigma, n: int, dtype=None, device=None):
    """SpectralGradient(σ, n) (:150-188): σI with a ONE-element `d`; T = typeof(σ) (Python float -> Float64)."""
    if not sigma > 0:
        raise AssertionError("σ > 0")                    # @assert σ > 0 (:185)
    if dtype is None:
        dtype = torch.float32 if isinstance(sigma, np.float32) else torch.float64
    dev = indexed_device(device)
    d = torch.tensor([float(sigma)], dtype=dtype, device=dev)
    return SpectralGradientType(d, n)


def push_diag(op: _DiagonalQN, s: torch.Tensor, y: torch.Tensor):
    return op._push(s, y)
