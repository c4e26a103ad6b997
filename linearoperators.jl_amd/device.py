"""Device plumbing: one ``mxlo_ctx`` per GPU per process, torch tensors as the device-vector type.

PyTorch is used here only for device memory (``torch.Tensor`` plays the role of the
Julia glue's device ``Vector{T}``), for the current HIP stream, and — in
:mod:`sharded` — for ``torch.distributed``. All arithmetic happens in libmxlo.so.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import torch

from . import _lib

_DT = {torch.float64: _lib.F64, torch.float32: _lib.F32}
_DTC = {torch.complex128: _lib.C64, torch.complex64: _lib.C32}


def dtype_code(dtype: torch.dtype, complex_ok: bool = False) -> int:
    """MXLO_F64 / MXLO_F32; with ``complex_ok`` also MXLO_C64 / MXLO_C32 (the elementwise leaves, opHouseholder,
    restriction/extension and the wrapper routing are instantiated for ComplexF64 / ComplexF32)."""
    if dtype in _DT:
        return _DT[dtype]
    if complex_ok and dtype in _DTC:
        return _DTC[dtype]
    if dtype in _DTC:
        raise TypeError(f"this leaf of the MI355X path is instantiated for float64/float32 only, got {dtype} "
                        "(complex: opDiagonal, opEye, opZeros, opHouseholder, restriction/extension, dense "
                        "LinearOperator(M), opHermitian and the adjoint/transpose/conj wrappers)")
    raise TypeError(f"the MI355X path is instantiated for float64/float32 (+ complex128/complex64 elementwise "
                    f"leaves) only, got {dtype} (BigFloat/Float16 stay on the reference CPU path)")


@dataclass(frozen=True)
class Storage:
    """The reference's storage type ``S`` (e.g. ``Vector{Float64}``): here a 1-D contiguous
    torch tensor of ``dtype`` on ``device`` (src/abstract.jl:176-184)."""
    dtype: torch.dtype
    device: torch.device

    def undef(self, n: int) -> torch.Tensor:          # S(undef, n)
        return torch.empty(n, dtype=self.dtype, device=self.device)

    def zeros(self, n: int) -> torch.Tensor:          # fill!(S(undef, n), zero(T))
        return torch.zeros(n, dtype=self.dtype, device=self.device)

    def __repr__(self):
        return f"DeviceVector{{{str(self.dtype).replace('torch.', '')}}}@{self.device}"


def storage_of(t: torch.Tensor) -> Storage:
    return Storage(t.dtype, t.device)


class Context:
    """RAII wrapper of ``mxlo_ctx`` bound to one GPU; kernels run on torch's current stream."""

    def __init__(self, index: int):
        self.index = index
        self.handle = C.c_void_p()
        with torch.cuda.device(index):
            self._stream = _raw_stream(index)
            _lib.call("mxlo_ctx_create", index, C.c_void_p(self._stream), C.byref(self.handle))
        self._hook = None  # keep the CFUNCTYPE object alive

    def bind_stream(self):
        # kernels run on torch's CURRENT stream of this device; the raw-handle query is ~10x cheaper than building a
        # torch.cuda.Stream object per apply (this runs on every mul!)
        s = _raw_stream(self.index)
        if s != self._stream:
            _lib.call("mxlo_ctx_set_stream", self.handle, C.c_void_p(s))
            self._stream = s

    @property
    def stream(self) -> int:
        """Raw hipStream_t the ctx currently launches on (0 = the null stream)."""
        return int(self._stream or 0)

    def sync(self):
        _lib.call("mxlo_ctx_sync", self.handle)

    def tune(self, key: str, value: int):
        _lib.call("mxlo_ctx_tune", self.handle, key.encode(), int(value))

    def info(self):
        a = (C.c_int64 * 4)()
        _lib.call("mxlo_ctx_info", self.handle, a)
        return {"device": a[0], "num_cu": a[1], "workspace_bytes": a[2], "max_reduction_cols": a[3]}

    def set_allreduce(self, pyfunc):
        """Install (or clear with None) the row-sharding all-reduce hook (include/mxlo.h)."""
        if pyfunc is None:
            self._hook = None
            _lib.call("mxlo_ctx_set_allreduce", self.handle, _lib.ALLREDUCE_FN(), None)
        else:
            self._hook = _lib.ALLREDUCE_FN(pyfunc)
            _lib.call("mxlo_ctx_set_allreduce", self.handle, self._hook, None)

    def __del__(self):
        try:
            if self.handle:
                _lib.lib().mxlo_ctx_destroy(self.handle)
        except Exception:
            pass


_ctxs: dict[int, Context] = {}
_available = None

try:                                                   # raw hipStream_t of torch's current stream (no Stream object)
    _raw_stream = torch._C._cuda_getCurrentRawStream
except AttributeError:                                 # pragma: no cover
    def _raw_stream(index: int) -> int:
        return torch.cuda.current_stream(index).cuda_stream


def get_ctx(device=None) -> Context:
    global _available
    if _available is None:
        _available = torch.cuda.is_available()
    if not _available:
        raise RuntimeError("no HIP device visible: the MI355X path has no CPU fallback")
    if device is None:
        idx = torch.cuda.current_device()
    else:
        if not isinstance(device, torch.device):
            device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError(f"operands must live on a GPU, got {device}: there is no CPU fallback")
        idx = device.index if device.index is not None else torch.cuda.current_device()
    ctx = _ctxs.get(idx)
    if ctx is None:
        ctx = _ctxs[idx] = Context(idx)
    ctx.bind_stream()
    return ctx


def indexed_device(device=None) -> torch.device:
    """`device` as an INDEXED cuda device: 'cuda' / torch.device('cuda') name the current device but compare unequal to
    the `cuda:0` every tensor reports, which would silently send `res.device == dev` checks down their slow branch."""
    if device is None:
        return torch.device("cuda", torch.cuda.current_device())
    if not isinstance(device, torch.device):
        device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        return torch.device("cuda", torch.cuda.current_device())
    return device


def ctx_of(t: torch.Tensor) -> Context:
    """`get_ctx(t.device)` for a tensor already known to live on a GPU (the per-apply path of the leaves): the device
    index comes from `get_device()` (no torch.device object), the ctx from the cache, and the stream is re-bound."""
    ctx = _ctxs.get(t.get_device())
    if ctx is None:
        return get_ctx(t.device)
    ctx.bind_stream()
    return ctx


def ptr(t: torch.Tensor | None):
    """The device address as a plain int (None for a NULL pointer): every entry point has `c_void_p` argtypes, which
    convert both — building a `c_void_p` object per operand was a measurable part of a launch-bound apply."""
    return None if t is None else t.data_ptr()


def check_vec(t: torch.Tensor, name: str, dtype: torch.dtype | None = None) -> torch.Tensor:
    """A Julia ``Vector``/contiguous ``SubArray``: 1-D, unit stride, on the GPU."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor on the GPU, got {type(t)}")
    if t.dim() != 1:
        raise ValueError(f"{name} must be 1-D, got shape {tuple(t.shape)}")
    if t.numel() > 1 and t.stride(0) != 1:
        raise ValueError(f"{name} must be contiguous (unit stride)")
    if not t.is_cuda:
        raise RuntimeError(f"{name} lives on {t.device}: the MI355X path has no CPU fallback")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name} has dtype {t.dtype}, expected {dtype}")
    return t


class Timer:
    """hipEvent pair recorded on the ctx stream (mxlo_timer_*)."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self.h = C.c_void_p()
        _lib.call("mxlo_timer_create", ctx.handle, C.byref(self.h))

    def start(self):
        _lib.call("mxlo_timer_start", self.h)

    def stop(self):
        _lib.call("mxlo_timer_stop", self.h)

    def elapsed_ms(self) -> float:
        ms = C.c_double()
        _lib.call("mxlo_timer_elapsed_ms", self.h, C.byref(ms))
        return ms.value

    def __del__(self):
        try:
            _lib.lib().mxlo_timer_destroy(self.h)
        except Exception:
            pass
