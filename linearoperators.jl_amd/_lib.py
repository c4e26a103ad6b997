"""ctypes binding of ``libmxlo.so`` (the C ABI declared in ``include/mxlo.h``).

This is the *only* compute path of the package: if the shared library is missing
or a call fails, an exception is raised — there is no CPU / PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import re
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("MXLO_LIB_PATH") or os.path.join(CSRC, "libmxlo.so")   # override: kernel experiments
RCCL_LIB_PATH = os.path.join(CSRC, "libmxlo_rccl.so")
HEADER = os.path.normpath(os.path.join(_HERE, "..", "include", "mxlo.h"))
RCCL_HEADER = os.path.normpath(os.path.join(_HERE, "..", "include", "mxlo_rccl.h"))
RCCL_ID_BYTES = 128
SHARD_AUTO, SHARD_RCCL, SHARD_LOOPBACK, SHARD_PEER = 0, 1, 2, 3

# status codes (include/mxlo.h)
OK, EINVAL, ESHAPE, EHIP, ENOMEM, ESTATE, EDOMAIN, EREDUCE = range(8)
F64, F32, C64, C32 = 0, 1, 2, 3
SHARD_PACK, SHARD_UNPACK = 0, 1
CONJ_D, ALPHA_REAL, BETA_REAL, D_REAL = 0x10, 0x20, 0x40, 0x80
ALPHA_F64, D_SCALAR, TAIL_BETA, BETA_F64 = 0x1, 0x2, 0x4, 0x8
SCALARS_F64 = ALPHA_F64 | BETA_F64
OP_N, OP_T, OP_C = 0, 1, 2
OP_J = 3
BLK_DIAG, BLK_DENSE, BLK_EYE, BLK_ZEROS, BLK_CSC = 0, 1, 2, 3, 4
QN_LBFGS_INV, QN_LBFGS_FWD, QN_LSR1 = 0, 1, 2
INV_TWOPASS, INV_REFORDER = 0, 1
PUSH_GRAM, PUSH_REFORDER, PUSH_COMPACT = 0, 1, 2
DQN_PSB, DQN_ANDREI, DQN_BFGS, DQN_SPECTRAL = 0, 1, 2, 3


class MxloError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"libmxlo status {status}: {msg}")
        self.status = status


class BlockDesc(C.Structure):
    """``mxlo_block_desc`` (include/mxlo.h)."""
    _fields_ = [("kind", C.c_int32), ("reserved", C.c_int32), ("row_off", C.c_int64), ("col_off", C.c_int64),
                ("m", C.c_int64), ("n", C.c_int64), ("data", C.c_void_p), ("ld", C.c_int64)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP translation unit for gfx950 and link ``libmxlo.so`` in-tree."""
    cmd = ["make", "-C", CSRC, "-j", str(max(2, (os.cpu_count() or 4))), "all"]
    if force:
        cmd.insert(1, "-B")
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or out.returncode != 0:
        print(out.stdout)
    if out.returncode != 0:
        raise RuntimeError("building libmxlo.so failed")
    return LIB_PATH


def header_symbols(header: str = None) -> list[str]:
    """Every function name declared in include/mxlo.h (or another header; used by the symbol-export test)."""
    text = open(header or HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(mxlo_[a-z0-9_]+)\s*\(", text)
    skip = {"mxlo_allreduce_fn"}
    out = []
    for n in names:
        if n not in skip and n not in out:
            out.append(n)
    return out


_lib = None

_vp, _i32, _i64, _dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
_PROTOS = {
    "mxlo_index_plan_create": [_vp, _vp, _i64, _i64, C.POINTER(_vp)],
    "mxlo_index_plan_destroy": [_vp],
    "mxlo_gather_plan": [_vp, _i32, _vp, _vp, _i64, _vp],
    "mxlo_scatter_zero_plan": [_vp, _i32, _vp, _i64, _vp, _vp, _vp],
    "mxlo_ctx_create": [_i32, _vp, C.POINTER(_vp)],
    "mxlo_ctx_destroy": [_vp],
    "mxlo_ctx_set_stream": [_vp, _vp],
    "mxlo_ctx_sync": [_vp],
    "mxlo_ctx_info": [_vp, C.POINTER(_i64)],
    "mxlo_ctx_tune": [_vp, C.c_char_p, _i64],
    "mxlo_ctx_set_allreduce": [_vp, ALLREDUCE_FN, _vp],
    "mxlo_malloc": [_vp, _i64, C.POINTER(_vp)],
    "mxlo_free": [_vp, _vp],
    "mxlo_memcpy_h2d": [_vp, _vp, _vp, _i64],
    "mxlo_memcpy_d2h": [_vp, _vp, _vp, _i64],
    "mxlo_memcpy_d2d": [_vp, _vp, _vp, _i64],
    "mxlo_memset": [_vp, _vp, _i32, _i64],
    "mxlo_ctx_create_stream": [_vp, C.POINTER(_vp)],
    "mxlo_graph_begin": [_vp],
    "mxlo_graph_end": [_vp, C.POINTER(_vp)],
    "mxlo_graph_launch": [_vp],
    "mxlo_graph_info": [_vp, C.POINTER(_i64)],
    "mxlo_debug_counters": [C.POINTER(_i64)],
    "mxlo_graph_destroy": [_vp],
    "mxlo_timer_create": [_vp, C.POINTER(_vp)],
    "mxlo_timer_start": [_vp],
    "mxlo_timer_stop": [_vp],
    "mxlo_timer_elapsed_ms": [_vp, C.POINTER(_dbl)],
    "mxlo_timer_destroy": [_vp],
    "mxlo_diag_mul": [_vp, _i32, _vp, _vp, _vp, _i64, _i64, _dbl, _dbl, _i32],
    "mxlo_eye_mul": [_vp, _i32, _vp, _vp, _i64, _i64, _dbl, _dbl, _i32],
    "mxlo_zeros_mul": [_vp, _i32, _vp, _i64, _dbl, _i32],
    "mxlo_ones_mul": [_vp, _i32, _vp, _i64, _vp, _i64, _dbl, _dbl, _i32],
    "mxlo_scale": [_vp, _i32, _vp, _i64, _dbl, _i32],
    "mxlo_fill": [_vp, _i32, _vp, _i64, _dbl],
    "mxlo_diag_mul_c": [_vp, _i32, _vp, _vp, _vp, _i64, _i64, _dbl, _dbl, _dbl, _dbl, _i32],
    "mxlo_eye_mul_c": [_vp, _i32, _vp, _vp, _i64, _i64, _dbl, _dbl, _dbl, _dbl, _i32],
    "mxlo_zeros_mul_c": [_vp, _i32, _vp, _i64, _dbl, _dbl, _i32],
    "mxlo_scale_c": [_vp, _i32, _vp, _i64, _dbl, _dbl, _i32],
    "mxlo_conj_c": [_vp, _i32, _vp, _vp, _i64],
    "mxlo_dot_c": [_vp, _i32, _vp, _vp, _i64, _vp],
    "mxlo_householder_mul_c": [_vp, _i32, _vp, _vp, _vp, _i64, _dbl, _dbl, _dbl, _dbl, _i32],
    "mxlo_kron_mul_c": [_vp, _i32, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _vp, _vp, _i64, _i64, _i64, _i32, _vp, _vp, _dbl, _dbl, _dbl, _dbl, _i32],
    "mxlo_kron_mul_c3": [_vp, _i32, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _vp, _vp, _dbl, _dbl, _dbl, _dbl, _i32],
    "mxlo_plane_sum": [_vp, _i32, _vp, _vp, _vp, _i64, _i64, _i64, _dbl],
    "mxlo_kron_c3_work_size": [_i64, _i64, _i32, _i64, _i64, _i32],
    "mxlo_split_c": [_vp, _i32, _vp, _vp, _vp, _i64],
    "mxlo_join_c": [_vp, _i32, _vp, _vp, _vp, _i64, _dbl, _dbl, _dbl, _dbl, _i32],
    "mxlo_gemv_c": [_vp, _i32, _vp, _vp, _i64, _i64, _i64, _vp, _dbl, _dbl, _dbl, _dbl, _i32, _i32],
    "mxlo_hermitian_mul_c": [_vp, _i32, _vp, _vp, _vp, _i64, _vp, _i64, _dbl, _dbl, _dbl, _dbl, _i32],
    "mxlo_householder_mul": [_vp, _i32, _vp, _vp, _vp, _i64, _dbl, _dbl, _i32],
    "mxlo_dot": [_vp, _i32, _vp, _vp, _i64, _vp],
    "mxlo_householder_apply": [_vp, _i32, _vp, _vp, _vp, _i64, _dbl, _dbl, _i32, _vp],
    "mxlo_hermitian_mul": [_vp, _i32, _vp, _vp, _vp, _i64, _vp, _i64, _dbl, _dbl, _i32],
    "mxlo_hermitian_mul_block": [_vp, _i32, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _i64, _i64, _dbl, _dbl, _i32],
    "mxlo_gather": [_vp, _i32, _vp, _vp, _i64, _vp, _i64],
    "mxlo_gather_range": [_vp, _i32, _vp, _vp, _i64, _i64, _i64, _i64],
    "mxlo_scatter_zero": [_vp, _i32, _vp, _i64, _vp, _vp, _vp, _i64],
    "mxlo_scatter_zero_range": [_vp, _i32, _vp, _i64, _vp, _i64, _i64, _i64],
    "mxlo_scatter_zero_sorted": [_vp, _i32, _vp, _i64, _vp, _vp, _vp, _i64],
    "mxlo_shard_stage": [_vp, _i32, _vp, _vp, _i64, _i32, _i64, _i32],
    "mxlo_blockdiag_create": [_vp, _i32, C.POINTER(BlockDesc), _i64, C.POINTER(_vp)],
    "mxlo_blockdiag_mul": [_vp, _vp, _vp, _dbl, _dbl, _i32, _i32],
    "mxlo_blockdiag_destroy": [_vp],
    "mxlo_kron_mul": [_vp, _i32, _vp, _vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _dbl, _dbl, _i32, _i32],
    "mxlo_kron_mul_ex": [_vp, _i32, _vp, _vp, _i64, _i64, _i64, _i32, _vp, _i64, _i64, _i64, _i32, _vp, _vp, _dbl, _dbl, _i32],
    "mxlo_kron_diag_mul": [_vp, _i32, _vp, _vp, _i64, _vp, _i64, _vp, _dbl, _dbl, _i32],
    "mxlo_gemv": [_vp, _i32, _vp, _vp, _i64, _i64, _i64, _vp, _dbl, _dbl, _i32, _i32],
    "mxlo_csc_create": [_vp, _i32, _i64, _i64, _vp, _vp, _vp, _i32, C.POINTER(_vp)],
    "mxlo_csc_refresh": [_vp],
    "mxlo_csc_mul": [_vp, _vp, _vp, _dbl, _dbl, _i32, _i32],
    "mxlo_csc_mul_c": [_vp, _vp, _vp, _dbl, _dbl, _dbl, _dbl, _i32, _i32],
    "mxlo_csc_mul_block": [_vp, _vp, _i64, _vp, _i64, _i64, _dbl, _dbl, _i32, _i32],
    "mxlo_csc_info": [_vp, C.POINTER(_i64)],
    "mxlo_debug_csc_chunks": [C.POINTER(_i64), _i64, C.POINTER(_i64), _i64, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)],
    "mxlo_csc_destroy": [_vp],
    "mxlo_gemv_block": [_vp, _i32, _vp, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _i64, _dbl, _dbl, _i32, _i32],
    "mxlo_diagqn_push": [_vp, _i32, _i32, _vp, _vp, _vp, _i64, C.POINTER(_i32)],
    "mxlo_qn_create": [_vp, _i32, _i32, _i64, _i64, _i32, _i32, _dbl, _dbl, C.POINTER(_vp)],
    "mxlo_qn_destroy": [_vp],
    "mxlo_qn_push": [_vp, _vp, _vp, C.POINTER(_i32)],
    "mxlo_qn_push_damped_fwd": [_vp, _vp, _vp, _vp, C.POINTER(_i32)],
    "mxlo_qn_push_damped_inv": [_vp, _vp, _vp, _dbl, _vp, _vp, C.POINTER(_i32)],
    "mxlo_qn_mul": [_vp, _vp, _vp, _dbl, _dbl, _i32],
    "mxlo_qn_mul_shifted": [_vp, _vp, _vp, _dbl, _dbl, _dbl, _i32],
    "mxlo_qn_solve_shifted": [_vp, _vp, _vp, _dbl],
    "mxlo_qn_diag": [_vp, _vp],
    "mxlo_qn_reset": [_vp],
    "mxlo_qn_get_scalars": [_vp, C.POINTER(_dbl), C.POINTER(_dbl), C.POINTER(_dbl)],
    "mxlo_qn_column": [_vp, _i32, _i64, C.POINTER(_vp)],
    "mxlo_qn_set_mode": [_vp, _i32],
    "mxlo_qn_set_push_mode": [_vp, _i32],
}


def lib() -> C.CDLL:
    """Load libmxlo.so. torch must be imported first so that the HIP runtime both sides
    use is the single ``libamdhip64.so.7`` already mapped into the process."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH) and os.path.exists("/opt/rocm/bin/hipcc"):
        # a source-only checkout (the .so is git-ignored): compile the HIP sources in-tree once. This is the
        # same library, not a fallback — if hipcc is absent or the build fails we raise below.
        try:
            build()
        except Exception as e:  # pragma: no cover
            raise ImportError(f"{LIB_PATH} is missing and building it failed: {e}") from e
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no fallback path.")
    import torch  # noqa: F401  (maps libamdhip64.so.7 first)
    L = C.CDLL(LIB_PATH)
    for name in ("mxlo_version", "mxlo_status_string", "mxlo_last_error"):
        getattr(L, name).restype = C.c_char_p
    L.mxlo_status_string.argtypes = [_i32]
    for name, args in _PROTOS.items():
        try:
            f = getattr(L, name)
        except AttributeError:
            continue  # symbol-export test reports it; calls through it raise below
        f.argtypes = args
        f.restype = _i64 if name == "mxlo_kron_c3_work_size" else _i32
    _lib = L
    return L


_rccl = None


def rccl_lib() -> C.CDLL:
    """libmxlo_rccl.so: the native RCCL transport of the all-reduce hook (include/mxlo_rccl.h)."""
    global _rccl
    if _rccl is None:
        lib()                                      # torch first (maps its librccl.so.1 / libamdhip64.so.7)
        if not os.path.exists(RCCL_LIB_PATH):
            raise ImportError(f"{RCCL_LIB_PATH} is missing: run __graft_entry__.build()")
        R = C.CDLL(RCCL_LIB_PATH)
        R.mxlo_rccl_unique_id.argtypes = [_vp]
        R.mxlo_rccl_comm_create.argtypes = [_i32, _i32, _vp, C.POINTER(_vp)]
        R.mxlo_rccl_comm_destroy.argtypes = [_vp]
        R.mxlo_rccl_allreduce_hook.argtypes = [_vp, _vp, _i64, _vp]
        R.mxlo_rccl_comm_abort.argtypes = [_vp]
        R.mxlo_rccl_comm_info.argtypes = [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32), C.c_char_p, _i32]
        R.mxlo_rccl_preflight.argtypes = [_vp, _vp, _i32, _i32, C.POINTER(_dbl)]
        R.mxlo_rccl_preflight_hook.argtypes = [_vp, _vp, _vp, _i32, _i32, _vp, _i32, _i32, C.POINTER(_dbl)]
        R.mxlo_peer_comm_create_shm.argtypes = [C.c_char_p, _i32, _i32, _i32, _i32, C.POINTER(_vp)]
        R.mxlo_peer_comm_destroy.argtypes = [_vp]
        R.mxlo_peer_allreduce_hook.argtypes = [_vp, _vp, _i64, _vp]
        R.mxlo_peer_comm_check.argtypes = [_vp]
        R.mxlo_peer_comm_debug.argtypes = [_vp, C.c_char_p, _i64]
        for f in (R.mxlo_rccl_unique_id, R.mxlo_rccl_comm_create, R.mxlo_rccl_comm_destroy, R.mxlo_rccl_allreduce_hook,
                  R.mxlo_rccl_comm_abort, R.mxlo_rccl_comm_info, R.mxlo_rccl_preflight, R.mxlo_rccl_preflight_hook,
                  R.mxlo_peer_comm_create_shm, R.mxlo_peer_comm_destroy, R.mxlo_peer_allreduce_hook, R.mxlo_peer_comm_check,
                  R.mxlo_peer_comm_debug):
            f.restype = _i32
        R.mxlo_rccl_last_error.restype = C.c_char_p
        R.mxlo_peer_last_error.restype = C.c_char_p
        # single-process multi-device API (include/mxlo_rccl.h)
        pp, ip = C.POINTER(_vp), C.POINTER(_i64)
        for name, args in {
            "mxlo_shard_ctx_create": [_i32, C.POINTER(_i32), C.POINTER(_vp)],
            "mxlo_shard_ctx_create_ex": [_i32, C.POINTER(_i32), _i32, C.POINTER(_vp)],
            "mxlo_shard_ctx_transport": [_vp],
            "mxlo_shard_ctx_info": [_vp, _i32, C.POINTER(_i32), C.POINTER(_i32), C.c_char_p, _i32],
            "mxlo_shard_ctx_preflight": [_vp, _i32, _i32, C.POINTER(_dbl)],
            "mxlo_shard_ctx_debug": [_vp, C.c_char_p, _i64],
            "mxlo_shard_ctx_destroy": [_vp], "mxlo_shard_ctx_ndev": [_vp], "mxlo_shard_ctx_device": [_vp, _i32],
            "mxlo_shard_ctx_is_loopback": [_vp], "mxlo_shard_ctx_sync": [_vp],
            "mxlo_householder_mul_sharded": [_vp, _i32, pp, pp, pp, ip, _dbl, _dbl, _i32],
            "mxlo_diag_mul_sharded": [_vp, _i32, pp, pp, pp, ip, _dbl, _dbl, _i32],
            "mxlo_qn_create_sharded": [_vp, _i32, _i32, ip, _i64, _i32, _i32, _dbl, _dbl, C.POINTER(_vp)],
            "mxlo_qn_destroy_sharded": [_vp],
            "mxlo_qn_push_sharded": [_vp, pp, pp, C.POINTER(_i32)],
            "mxlo_qn_mul_sharded": [_vp, pp, pp, _dbl, _dbl, _i32],
            "mxlo_qn_mul_shifted_sharded": [_vp, pp, pp, _dbl, _dbl, _dbl, _i32],
            "mxlo_qn_solve_shifted_sharded": [_vp, pp, pp, _dbl],
            "mxlo_qn_diag_sharded": [_vp, pp],
            "mxlo_qn_reset_sharded": [_vp],
            "mxlo_qn_get_scalars_sharded": [_vp, _i32, C.POINTER(_dbl), C.POINTER(_dbl), C.POINTER(_dbl)],
        }.items():
            f = getattr(R, name)
            f.argtypes = args
            f.restype = _i32
        R.mxlo_shard_ctx_get.argtypes = [_vp, _i32]
        R.mxlo_shard_ctx_get.restype = _vp
        R.mxlo_qn_sharded_get.argtypes = [_vp, _i32]
        R.mxlo_qn_sharded_get.restype = _vp
        R.mxlo_shard_last_error.restype = C.c_char_p
        _rccl = R
    return _rccl


def check(status: int) -> None:
    if status != OK:
        L = lib()
        msg = (L.mxlo_last_error() or b"").decode() or L.mxlo_status_string(status).decode()
        raise MxloError(status, msg)


_FN: dict = {}


def call(name: str, *args) -> None:
    f = _FN.get(name)
    if f is None:
        try:
            f = _FN[name] = getattr(lib(), name)
        except AttributeError as e:  # pragma: no cover
            raise ImportError(f"libmxlo.so does not export {name}; rebuild it") from e
    st = f(*args)
    if st != OK:
        check(st)
