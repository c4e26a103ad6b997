"""CPU oracle for the LinearOperators.jl `mul!` hot path — TEST INFRASTRUCTURE ONLY.

ctypes front-end of ``liblo_oracle.so`` (``lo_oracle.c`` / ``lo_oracle_impl.h``:
a statement-by-statement C restatement of the reference closures, each function
citing the ``src/<file>.jl:<line>`` it follows) plus a small NumPy restatement of
the reference's *host* logic (5-arg ``mul!`` dispatch, ``prod3!``, combinators,
cat) in :mod:`oracle.refops`.

Pinning: Julia is not installed in this image, so the reference itself cannot be
run; the oracle is pinned against the known-answer cases held by the reference's
own tests (``tests/golden/kat_reference_tests.json``) and against independent
dense-matrix constructions, the way the reference's tests pin the package.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package. The product never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liblo_oracle.so")

ALPHA_F64 = 0x1
BETA_F64 = 0x8
SCALARS_F64 = ALPHA_F64 | BETA_F64


CONJ_D = 0x10
ALPHA_REAL = 0x20
BETA_REAL = 0x40


def _julia_f64(x) -> bool:
    """A Python float / complex or a float64 / complex128 scalar stands for a Julia Float64 / ComplexF64; Python
    ints are Julia Ints (they never widen a Float32 product); NumPy float32 / complex64 scalars are 32-bit."""
    if isinstance(x, (bool, int, np.integer)):
        return False
    if isinstance(x, np.floating):
        return x.dtype == np.float64
    if isinstance(x, np.complexfloating):
        return x.dtype == np.complex128
    return True


def _is_real_scalar(x) -> bool:
    return not isinstance(x, (complex, np.complexfloating))


def scalar_flags(dtype, alpha, beta) -> int:
    """ORC_ALPHA_F64 / ORC_BETA_F64 for Float32 / ComplexF32 data (Julia's mixed-precision rule) and
    ORC_ALPHA_REAL / ORC_BETA_REAL for Real scalars next to complex data."""
    dt = np.dtype(dtype)
    fl = 0
    if dt in (np.float32, np.complex64):
        fl |= (ALPHA_F64 if _julia_f64(alpha) else 0) | (BETA_F64 if _julia_f64(beta) else 0)
    if dt.kind == "c":
        fl |= (ALPHA_REAL if _is_real_scalar(alpha) else 0) | (BETA_REAL if _is_real_scalar(beta) else 0)
    return fl
D_SCALAR = 0x2
TAIL_BETA = 0x4


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (``make -C oracle``)."""
    src_m = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("lo_oracle.c", "lo_oracle_impl.h", "lo_oracle_cplx.h"))
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < src_m:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "all"])
    return _LIB_PATH


_MT_PATH = os.path.join(_HERE, "liblo_oracle_mt.so")
_mt = None


def mt_lib() -> C.CDLL:
    """All-core (OpenMP) timing variant, bench.py cpu_baseline upper bound only (lo_oracle_mt.c)."""
    global _mt
    if _mt is None:
        src = os.path.join(_HERE, "lo_oracle_mt.c")
        if not os.path.exists(_MT_PATH) or os.path.getmtime(_MT_PATH) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liblo_oracle_mt.so"])
        _mt = C.CDLL(_MT_PATH)
        _mt.orc_mt_fill_f64.argtypes = [C.c_void_p, C.c_int64, C.c_uint64, C.c_double, C.c_double, C.c_int32]
        _mt.orc_mt_householder_mul_f64.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_double,
                                                   C.c_double, C.c_int32]
        _mt.orc_mt_max_threads.restype = C.c_int32
        _mt.orc_mt_lbfgs_inv_mul_f64.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32,
                                                 C.c_int32, C.c_int32, C.c_double, C.c_void_p, C.c_void_p, C.c_int64, C.c_double,
                                                 C.c_double, C.c_int32]
        _mt.orc_mt_kron_mul_f64.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64,
                                            C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_int32]
    return _mt


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
    return _lib


def _suf(dtype) -> str:
    dtype = np.dtype(dtype)
    if dtype == np.float64:
        return "f64"
    if dtype == np.float32:
        return "f32"
    if dtype == np.complex128:
        return "c64"
    if dtype == np.complex64:
        return "c32"
    raise TypeError(f"oracle instantiated for float64/float32 only, got {dtype}")


def _p(a: np.ndarray):
    assert a.flags["C_CONTIGUOUS"] or a.flags["F_CONTIGUOUS"], "oracle needs contiguous arrays"
    return a.ctypes.data_as(C.c_void_p)


def _fn(name: str, dtype):
    return getattr(lib(), f"{name}_{_suf(dtype)}")


_d = C.c_double
_i64 = C.c_int64
_i32 = C.c_int32


def dotc(h, v):
    """LinearAlgebra.dot(h, v) for complex vectors (conjugates h)."""
    out = np.zeros(1, dtype=h.dtype)
    _fn("orc_dotc", h.dtype)(_p(h), _p(v), _i64(h.size), _p(out))
    return out[0]


def _check(res, *vs):
    dt = res.dtype
    for v in vs:
        assert v.dtype == dt, f"dtype mismatch {v.dtype} vs {dt}"
    return dt


# --------------------------------------------------------------------------- leaves
def dot(a, b):
    dt = _check(a, b)
    f = _fn("orc_dot_public", dt)
    f.restype = C.c_double if dt == np.float64 else C.c_float
    return dt.type(f(_p(a), _p(b), _i64(a.size)))


def _c4(alpha, beta):
    a, b = complex(alpha), complex(beta)
    return _d(a.real), _d(a.imag), _d(b.real), _d(b.imag)


def diag_mul(res, d, v, alpha, beta, n_min=None, flags=0):
    """mulSquareOpDiagonal!/mulOpDiagonal! (src/special-operators.jl:125-151). Complex arrays: `flags` carries
    CONJ_D for the ctprod! form and the scalar kinds (use scalar_flags)."""
    dt = _check(res, d, v)
    nrow = res.size
    n_min = nrow if n_min is None else n_min
    if dt.kind == "c":
        _fn("orc_diag_mul", dt)(_p(res), _p(d), _p(v), _i64(n_min), _i64(nrow), *_c4(alpha, beta), _i32(flags))
        return res
    _fn("orc_diag_mul", dt)(_p(res), _p(d), _p(v), _i64(n_min), _i64(nrow), _d(alpha), _d(beta), _i32(flags))
    return res


def eye_mul(res, v, alpha, beta, n_min=None, flags=TAIL_BETA):
    """mulOpEye! (src/special-operators.jl:36-44)."""
    dt = _check(res, v)
    nrow = res.size
    n_min = min(nrow, v.size) if n_min is None else n_min
    if dt.kind == "c":
        _fn("orc_eye_mul", dt)(_p(res), _p(v), _i64(n_min), _i64(nrow), *_c4(alpha, beta), _i32(flags))
        return res
    _fn("orc_eye_mul", dt)(_p(res), _p(v), _i64(n_min), _i64(nrow), _d(alpha), _d(beta), _i32(flags))
    return res


def zeros_mul(res, beta, flags=0):
    if res.dtype.kind == "c":   # res .= 0 | res .*= β
        b = complex(beta)
        if b == 0:
            res[:] = 0
        else:
            _fn("orc_scale", res.dtype)(_p(res), _i64(res.size), _d(b.real), _d(b.imag), _i32(1 if flags & BETA_REAL else 0),
                                         _i32(1 if flags & BETA_F64 else 0))
        return res
    _fn("orc_zeros_mul", res.dtype)(_p(res), _i64(res.size), _d(beta), _i32(flags))
    return res


def ones_mul(res, v, alpha, beta, flags=0):
    dt = _check(res, v)
    _fn("orc_ones_mul", dt)(_p(res), _i64(res.size), _p(v), _i64(v.size), _d(alpha), _d(beta), _i32(flags))
    return res


def scale(res, alpha, flags=0):
    if res.dtype.kind == "c":
        a = complex(alpha)
        _fn("orc_scale", res.dtype)(_p(res), _i64(res.size), _d(a.real), _d(a.imag), _i32(1 if flags & ALPHA_REAL else 0),
                                     _i32(1 if flags & ALPHA_F64 else 0))
        return res
    _fn("orc_scale", res.dtype)(_p(res), _i64(res.size), _d(alpha), _i32(flags))
    return res


def householder_mul(res, h, v, alpha, beta, flags=0):
    if res.dtype.kind == "c":
        dt = _check(res, h, v)
        _fn("orc_householder_mul", dt)(_p(res), _p(h), _p(v), _i64(res.size), *_c4(alpha, beta), _i32(flags))
        return res
    """mulHouseholder! (src/linalg.jl:77-83)."""
    dt = _check(res, h, v)
    _fn("orc_householder_mul", dt)(_p(res), _p(h), _p(v), _i64(res.size), _d(alpha), _d(beta), _i32(flags))
    return res


def hermitian_mul(res, d, A, v, alpha, beta, flags=0):
    """mulHermitian! with L = tril(A,-1) (src/linalg.jl:97-116). A: (n,n) array, any order. Complex res: d real or complex."""
    n = res.size
    if res.dtype.kind == "c":
        dt = res.dtype
        assert v.dtype == dt
        rdt = np.float64 if dt == np.complex128 else np.float32
        d_real = d.dtype.kind != "c"
        dd = np.ascontiguousarray(d, dtype=rdt if d_real else dt)
        Af = np.asfortranarray(A, dtype=dt)
        t1, t2 = np.empty(n, dt), np.empty(n, dt)
        _fn("orc_hermitian_mul", dt)(_p(res), _p(dd), _i32(int(d_real)), _p(Af), _i64(n), _p(v), _i64(n), *_c4(alpha, beta),
                                     _i32(flags), _p(t1), _p(t2))
        return res
    dt = _check(res, d, v)
    Af = np.asfortranarray(A, dtype=dt)
    t1 = np.empty(n, dt)
    t2 = np.empty(n, dt)
    _fn("orc_hermitian_mul", dt)(_p(res), _p(d), _p(Af), _i64(n), _p(v), _i64(n), _d(alpha), _d(beta), _i32(flags), _p(t1), _p(t2))
    return res


def restrict(res, v, idx):
    """mulRestrict! (src/special-operators.jl:167-169); idx 1-based int64."""
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    es = res.dtype.itemsize
    assert res.dtype == v.dtype
    lib().orc_restrict_bytes(_p(res), _p(v), _p(idx), _i64(idx.size), _i64(es))
    return res


def extend(res, u, idx):
    """multRestrict! (src/special-operators.jl:171-174); idx 1-based int64."""
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    es = res.dtype.itemsize
    assert res.dtype == u.dtype
    lib().orc_extend_bytes(_p(res), _i64(res.size), _p(u), _p(idx), _i64(idx.size), _i64(es))
    return res


def gemv(res, M, v, alpha, beta, trans=False, flags=0):
    """mul!(res, op(M), v, α, β) of a dense LinearOperator(M) (src/constructors.jl:19-29). Real data: trans False/True.
    Complex data: trans in {False|"N", True|"T", "C" (adjoint), "J" (conj(M)*v)}."""
    if res.dtype.kind == "c":
        dt = res.dtype
        assert v.dtype == dt
        mode = {False: 0, True: 1, "N": 0, "T": 1, "C": 2, "J": 3}[trans]
        Mf = np.asfortranarray(M, dtype=dt)
        m, n = Mf.shape
        tmp = np.empty(res.size, dt)
        _fn("orc_gemv", dt)(_p(res), _p(Mf), _i64(m), _i64(n), _i64(m), _p(v), *_c4(alpha, beta), _i32(mode), _i32(flags), _p(tmp))
        return res
    dt = _check(res, v)
    Mf = np.asfortranarray(M, dtype=dt)
    m, n = Mf.shape
    tmp = np.empty(res.size, dt)
    _fn("orc_gemv", dt)(_p(res), _p(Mf), _i64(m), _i64(n), _i64(m), _p(v), _d(alpha), _d(beta), _i32(int(trans)), _i32(flags), _p(tmp))
    return res


def csc_mul(res, colptr, rowval, nzval, m, n, v, alpha, beta, trans=False, flags=0):
    """mul!(res, op(A), v, α, β) for A::SparseMatrixCSC (src/constructors.jl:19-29 -> SparseArrays `_spmatmul!` /
    `_At_or_Ac_mul_B!`). colptr / rowval 1-based int64 as Julia stores them; real element types."""
    cp = np.ascontiguousarray(colptr, dtype=np.int64)
    rv = np.ascontiguousarray(rowval, dtype=np.int64)
    if res.dtype.kind == "c":                    # complex: trans in {False, True | "T", "C" (adjoint)}
        dt = res.dtype
        assert v.dtype == dt
        mode = {False: 0, True: 1, "N": 0, "T": 1, "C": 2}[trans]
        nz = np.ascontiguousarray(nzval, dtype=dt)
        assert cp.size == n + 1 and res.size == (n if mode else m) and v.size == (m if mode else n)
        _fn("orc_csc_mul", dt)(_p(res), _p(cp), _p(rv), _p(nz), _i64(m), _i64(n), _p(v), *_c4(alpha, beta), _i32(mode), _i32(flags))
        return res
    dt = _check(res, v)
    nz = np.ascontiguousarray(nzval, dtype=dt)
    assert cp.size == n + 1 and res.size == (n if trans else m) and v.size == (m if trans else n)
    _fn("orc_csc_mul", dt)(_p(res), _p(cp), _p(rv), _p(nz), _i64(m), _i64(n), _p(v), _d(alpha), _d(beta), _i32(int(trans)),
                           _i32(flags))
    return res


def kron_mul(res, A, B, x, alpha, beta, trans=False, flags=0):
    """kron(A,B) prod!/tprod!/ctprod! (src/kron.jl:14-40). Complex res: trans in {False, True|"T", "C"}; real factors
    next to complex data are promoted (kron(Float64 A, ComplexF64 B), test/test_kron.jl:3-8)."""
    if res.dtype.kind == "c":
        dt = res.dtype
        assert x.dtype == dt
        mode = {False: 0, True: 1, "N": 0, "T": 1, "C": 2}[trans]
        Af, Bf = np.asfortranarray(A, dtype=dt), np.asfortranarray(B, dtype=dt)
        m, n = Af.shape
        p, q = Bf.shape
        nro, nco = (q, n) if mode else (p, m)
        xr = p if mode else q
        work = np.empty(xr + nro * nco, dt)
        _fn("orc_kron_mul", dt)(_p(res), _p(Af), _i64(m), _i64(n), _i64(m), _p(Bf), _i64(p), _i64(q), _i64(p), _p(x),
                                *_c4(alpha, beta), _i32(mode), _i32(flags), _p(work))
        return res
    dt = _check(res, x)
    Af = np.asfortranarray(A, dtype=dt)
    Bf = np.asfortranarray(B, dtype=dt)
    m, n = Af.shape
    p, q = Bf.shape
    nro, nco = (q, n) if trans else (p, m)
    xr = p if trans else q
    work = np.empty(xr + nro * nco, dt)
    _fn("orc_kron_mul", dt)(_p(res), _p(Af), _i64(m), _i64(n), _i64(m), _p(Bf), _i64(p), _i64(q), _i64(p), _p(x), _d(alpha), _d(beta), _i32(int(trans)), _i32(flags), _p(work))
    return res


# --------------------------------------------------------------------------- L-BFGS
def _lbfgs_struct(ct):
    class S(C.Structure):
        _fields_ = [
            ("n", C.c_int64), ("mem", C.c_int64),
            ("scaling", C.c_int32), ("damped", C.c_int32), ("inverse", C.c_int32), ("pad_", C.c_int32),
            ("scaling_factor", ct), ("sigma2", ct), ("sigma3", ct), ("opnorm_upper_bound", ct),
            ("s", C.c_void_p), ("y", C.c_void_p), ("ys", C.c_void_p), ("alpha", C.c_void_p),
            ("a", C.c_void_p), ("b", C.c_void_p), ("norm_b", C.c_void_p),
            ("insert", C.c_int64),
            ("Ax", C.c_void_p), ("shifted_p", C.c_void_p), ("shifted_v", C.c_void_p), ("shifted_u", C.c_void_p),
        ]
    return S


_LBFGS_S = {"f64": _lbfgs_struct(C.c_double), "f32": _lbfgs_struct(C.c_float)}


class LBFGS:
    """LBFGSData + LBFGSOperator / InverseLBFGSOperator (src/lbfgs.jl:4-206)."""

    def __init__(self, n, mem=5, scaling=True, damped=False, inverse=True, sigma2=0.99, sigma3=10.0, dtype=np.float64):
        dt = np.dtype(dtype)
        self.dtype, self.n = dt, n
        self.mem = mem
        self.inverse = inverse
        z = lambda *sh: np.zeros(sh, dt)
        self.s, self.y = z(mem, n), z(mem, n)
        self.ys = z(mem)
        self.alpha = z(mem)
        self.a, self.b = z(mem, n), z(mem, n)      # (allocated for both forms; forward only uses them)
        self.norm_b = z(mem)
        self.Ax = z(n)
        self.shifted_p = z(2 * mem, n)             # column i of the Julia n x 2mem matrix = row i here
        self.shifted_v = z(2 * mem)
        self.shifted_u = z(n)
        S = _LBFGS_S[_suf(dt)]
        self.st = S(n=n, mem=max(mem, 1), scaling=int(scaling), damped=int(damped), inverse=int(inverse),
                    scaling_factor=1, sigma2=dt.type(sigma2), sigma3=dt.type(sigma3), opnorm_upper_bound=1,
                    s=_p(self.s).value, y=_p(self.y).value, ys=_p(self.ys).value, alpha=_p(self.alpha).value,
                    a=_p(self.a).value, b=_p(self.b).value, norm_b=_p(self.norm_b).value, insert=1,
                    Ax=_p(self.Ax).value, shifted_p=_p(self.shifted_p).value, shifted_v=_p(self.shifted_v).value,
                    shifted_u=_p(self.shifted_u).value)
        self.damped = damped

    # fields tests read
    insert = property(lambda self: int(self.st.insert))
    scaling_factor = property(lambda self: float(self.st.scaling_factor))
    opnorm_upper_bound = property(lambda self: float(self.st.opnorm_upper_bound))

    def mul(self, res, x, alpha=1.0, beta=0.0, flags=0):
        name = "orc_lbfgs_inv_mul" if self.inverse else "orc_lbfgs_fwd_mul"
        _fn(name, self.dtype)(C.byref(self.st), _p(res), _p(x), _d(alpha), _d(beta), _i32(flags))
        return res

    def push(self, s, y, alpha=None, g=None, Bs=None):
        """push! variants (src/lbfgs.jl:269-367). Returns True when the pair was stored."""
        dt = self.dtype
        f = None
        if not self.damped:
            if Bs is not None or alpha is not None:
                raise RuntimeError("This push! should be used for damped operators")  # :295, :331
            f = _fn("orc_lbfgs_push", dt)
            f.restype = C.c_int32
            return bool(f(C.byref(self.st), _p(s), _p(y)))
        if self.inverse:
            if alpha is None:
                raise RuntimeError("This function be used for inverse operators. Use push!(op, s, y, Bs) instead.")
            Bs = np.empty_like(s) if Bs is None else Bs
            f = _fn("orc_lbfgs_push_damped_inv", dt)
            f.restype = C.c_int32
            ct = C.c_double if dt == np.float64 else C.c_float
            return bool(f(C.byref(self.st), _p(s), _p(y), ct(alpha), _p(g), _p(Bs)))
        if alpha is not None:
            raise RuntimeError("This function be used for forward operators. Use push!(op, s, y, α, g, Bs) instead.")
        Bs = np.empty_like(s) if Bs is None else Bs
        ytmp = np.empty_like(s)
        f = _fn("orc_lbfgs_push_damped_fwd", dt)
        f.restype = C.c_int32
        return bool(f(C.byref(self.st), _p(s), _p(y), _p(Bs), _p(ytmp)))

    def diag(self):
        if self.inverse:
            raise ValueError("only the diagonal of a forward L-BFGS approximation is available")
        d = np.empty(self.n, self.dtype)
        _fn("orc_lbfgs_diag", self.dtype)(C.byref(self.st), _p(d))
        return d

    def reset(self):
        _fn("orc_lbfgs_reset", self.dtype)(C.byref(self.st))

    def solve_shifted(self, x, b, sigma):
        f = _fn("orc_solve_shifted", self.dtype)
        f.restype = C.c_int32
        ct = C.c_double if self.dtype == np.float64 else C.c_float
        if f(C.byref(self.st), _p(x), _p(b), ct(sigma)) != 0:
            raise ValueError("σ must be nonnegative")
        return x

    def dense(self):
        """Matrix(op) (src/abstract.jl:282-292)."""
        n = self.n
        M = np.empty((n, n), self.dtype)
        e = np.zeros(n, self.dtype)
        col = np.empty(n, self.dtype)
        for i in range(n):
            e[i] = 1
            M[:, i] = self.mul(col, e)
            e[i] = 0
        return M


# --------------------------------------------------------------------------- L-SR1
def _lsr1_struct(ct):
    class S(C.Structure):
        _fields_ = [
            ("n", C.c_int64), ("mem", C.c_int64), ("scaling", C.c_int32), ("pad_", C.c_int32),
            ("scaling_factor", ct), ("opnorm_upper_bound", ct),
            ("s", C.c_void_p), ("y", C.c_void_p), ("ys", C.c_void_p), ("a", C.c_void_p), ("as_", C.c_void_p),
            ("insert", C.c_int64), ("Ax", C.c_void_p), ("tmp", C.c_void_p),
        ]
    return S


_LSR1_S = {"f64": _lsr1_struct(C.c_double), "f32": _lsr1_struct(C.c_float)}


class LSR1:
    """LSR1Data + LSR1Operator (src/lsr1.jl:4-111)."""

    def __init__(self, n, mem=5, scaling=True, dtype=np.float64):
        dt = np.dtype(dtype)
        self.dtype, self.n, self.mem = dt, n, mem
        z = lambda *sh: np.zeros(sh, dt)
        self.s, self.y, self.a = z(mem, n), z(mem, n), z(mem, n)
        self.ys, self.as_ = z(mem), z(mem)
        self.Ax, self.tmp = z(n), z(n)
        S = _LSR1_S[_suf(dt)]
        self.st = S(n=n, mem=max(mem, 1), scaling=int(scaling), scaling_factor=1, opnorm_upper_bound=1,
                    s=_p(self.s).value, y=_p(self.y).value, ys=_p(self.ys).value, a=_p(self.a).value,
                    as_=_p(self.as_).value, insert=1, Ax=_p(self.Ax).value, tmp=_p(self.tmp).value)

    insert = property(lambda self: int(self.st.insert))
    scaling_factor = property(lambda self: float(self.st.scaling_factor))
    opnorm_upper_bound = property(lambda self: float(self.st.opnorm_upper_bound))

    def mul(self, res, x, alpha=1.0, beta=0.0, flags=0):
        _fn("orc_lsr1_mul", self.dtype)(C.byref(self.st), _p(res), _p(x), _d(alpha), _d(beta), _i32(flags))
        return res

    def push(self, s, y):
        f = _fn("orc_lsr1_push", self.dtype)
        f.restype = C.c_int32
        return bool(f(C.byref(self.st), _p(s), _p(y)))

    def diag(self):
        d = np.empty(self.n, self.dtype)
        _fn("orc_lsr1_diag", self.dtype)(C.byref(self.st), _p(d))
        return d

    def reset(self):
        _fn("orc_lsr1_reset", self.dtype)(C.byref(self.st))

    def dense(self):
        n = self.n
        M = np.empty((n, n), self.dtype)
        e = np.zeros(n, self.dtype)
        col = np.empty(n, self.dtype)
        for i in range(n):
            e[i] = 1
            M[:, i] = self.mul(col, e)
            e[i] = 0
        return M


# ---------------------------------------------------------------------------------------------------------
# Diagonal quasi-Newton push! — src/DiagonalHessianApproximation.jl, statement by statement in NumPy (every
# scalar stays in the eltype T like the reference; `dot`/`norm` orders are unpinned there, np.dot here).
class DiagonalQN:
    """kind in {"psb", "andrei", "bfgs", "spectral"}; `d` is the operator's diagonal (1 element for spectral)."""

    def __init__(self, kind, d):
        self.kind, self.d = kind, np.array(d, copy=True)
        self.T = self.d.dtype.type

    def mul(self, res, v, alpha=1.0, beta=0.0, flags=0):
        """mulSquareOpDiagonal! on the operator's own d (:37,112,179,226)."""
        return diag_mul(res, self.d, v, alpha, beta, flags=flags | (D_SCALAR if self.d.size == 1 and v.size != 1 else 0))

    def push(self, s, y):
        T, d = self.T, self.d
        if self.kind == "spectral":                          # :190-199
            if np.all(s == 0):
                raise ZeroDivisionError("Cannot divide by zero and s .= 0")
            d[0] = T(np.dot(s, y)) / T(np.dot(s, s))
            return self
        sNorm = T(np.linalg.norm(s, 2))                      # :50,122,241
        if sNorm == 0:
            raise ZeroDivisionError("Cannot update DiagonalQN operator with s=0")
        sNorm2 = T(sNorm * sNorm)                            # sNorm^2
        sT_y = T(T(np.dot(s, y)) / sNorm2)
        if self.kind == "bfgs":                              # :245-248
            d[:] = np.abs(y)
            d *= T(T(np.sum(d)) / sT_y)
            return self
        s2 = s * s                                           # (si^2 for si in s)
        trA2 = T(T(np.dot(s2, s2)) / T(sNorm2 * sNorm2))     # :57,129
        sT_B_s = T(T(np.dot(s2, d)) / sNorm2)
        q = T(sT_y - sT_B_s)
        if self.kind == "andrei":                            # :133-134
            q = T(q + T(T(np.dot(s, s)) / sNorm2))
        q = T(q / trA2)
        c = T(q / sNorm2)
        if self.kind == "psb":
            d += c * s2                                      # B.d .+= q / sNorm2 .* s .^ 2        (:62)
        else:
            d += (c * s2) - T(1)                             # B.d .+= q / sNorm2 .* s .^ 2 .- 1   (:137)
        return self

    def reset(self):
        self.d[:] = 1
        return self
