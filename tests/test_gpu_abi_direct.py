"""-m gpu: entry points of include/mxlo.h that the operator-level tests reach only indirectly — called
straight through ctypes the way a foreign-language glue would: context / memory / timer helpers,
mxlo_dot + mxlo_householder_apply, error codes (no exception crosses the ABI), tune keys."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def test_ctx_memory_timer_roundtrip(lo, dev):
    L = lo._lib.lib()
    ctx = C.c_void_p()
    assert L.mxlo_ctx_create(0, None, C.byref(ctx)) == 0
    info = (C.c_int64 * 4)()
    assert L.mxlo_ctx_info(ctx, info) == 0 and info[1] >= 1 and info[3] >= 40
    n = 1000
    host = np.arange(n, dtype=np.float64)
    p = C.c_void_p()
    assert L.mxlo_malloc(ctx, n * 8, C.byref(p)) == 0 and p.value
    assert L.mxlo_memcpy_h2d(ctx, p, host.ctypes.data_as(C.c_void_p), n * 8) == 0
    q = C.c_void_p()
    assert L.mxlo_malloc(ctx, n * 8, C.byref(q)) == 0
    assert L.mxlo_memset(ctx, q, 0, n * 8) == 0
    # res = 2*p + 0*q through the ABI, timed with the ABI's own timer
    t = C.c_void_p()
    assert L.mxlo_timer_create(ctx, C.byref(t)) == 0
    assert L.mxlo_timer_start(t) == 0
    assert L.mxlo_eye_mul(ctx, lo._lib.F64, q, p, n, n, 2.0, 0.0, 0) == 0
    assert L.mxlo_timer_stop(t) == 0
    ms = C.c_double(-1)
    assert L.mxlo_timer_elapsed_ms(t, C.byref(ms)) == 0 and ms.value >= 0
    back = np.empty(n)
    assert L.mxlo_memcpy_d2h(ctx, back.ctypes.data_as(C.c_void_p), q, n * 8) == 0
    assert np.array_equal(back, 2 * host)
    assert L.mxlo_memcpy_d2d(ctx, p, q, n * 8) == 0 and L.mxlo_ctx_sync(ctx) == 0
    assert L.mxlo_memcpy_d2h(ctx, back.ctypes.data_as(C.c_void_p), p, n * 8) == 0
    assert np.array_equal(back, 2 * host)
    assert L.mxlo_timer_destroy(t) == 0 and L.mxlo_free(ctx, p) == 0 and L.mxlo_free(ctx, q) == 0
    assert L.mxlo_ctx_set_stream(ctx, None) == 0
    assert L.mxlo_ctx_destroy(ctx) == 0


def test_status_codes_not_exceptions(lo, dev):
    L = lo._lib.lib()
    ctx = lo.get_ctx(dev).handle
    x = torch.ones(8, dtype=torch.float64, device=dev)
    px = C.c_void_p(x.data_ptr())
    assert L.mxlo_diag_mul(None, 0, px, px, px, 8, 8, 1.0, 0.0, 0) == lo._lib.EINVAL
    assert L.mxlo_diag_mul(ctx, 7, px, px, px, 8, 8, 1.0, 0.0, 0) == lo._lib.EINVAL
    assert L.mxlo_diag_mul(ctx, 0, px, px, px, 9, 8, 1.0, 0.0, 0) == lo._lib.ESHAPE
    assert b"n_min" in L.mxlo_last_error()
    assert L.mxlo_gather_range(ctx, 8, px, px, 8, 1, 1, 9) == lo._lib.ESHAPE         # 1:9 of an 8-vector
    assert L.mxlo_gather(ctx, 3, px, px, 8, px, 1) == lo._lib.EINVAL                  # element size 3
    assert L.mxlo_ctx_tune(ctx, b"no_such_key", 1) == lo._lib.EINVAL
    h = C.c_void_p()
    assert L.mxlo_qn_create(ctx, lo._lib.QN_LBFGS_FWD, 0, 8, 5000, 1, 0, 0.99, 10.0, C.byref(h)) == lo._lib.EINVAL  # mem > 4096
    assert L.mxlo_qn_create(ctx, lo._lib.QN_LBFGS_INV, 0, 8, 3, 1, 0, 0.99, 10.0, C.byref(h)) == 0
    assert L.mxlo_qn_solve_shifted(h, px, px, 0.5) == lo._lib.ESTATE                   # inverse operator
    assert L.mxlo_qn_diag(h, px) == lo._lib.ESTATE
    acc = C.c_int32()
    assert L.mxlo_qn_push_damped_fwd(h, px, px, px, C.byref(acc)) == lo._lib.ESTATE    # not damped
    assert L.mxlo_qn_destroy(h) == 0
    assert L.mxlo_qn_create(ctx, lo._lib.QN_LBFGS_FWD, 0, 8, 3, 1, 0, 0.99, 10.0, C.byref(h)) == 0
    assert L.mxlo_qn_solve_shifted(h, px, px, -0.1) == lo._lib.EDOMAIN                 # ArgumentError in the reference
    assert L.mxlo_qn_destroy(h) == 0
    assert L.mxlo_status_string(lo._lib.EDOMAIN) == b"argument outside domain"


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_dot_and_householder_apply(lo, dev, dtype):
    from linearoperators_jl_amd.device import dtype_code, get_ctx, ptr
    rng = np.random.default_rng(4)
    npd = np.float64 if dtype == torch.float64 else np.float32
    ctx = get_ctx(dev)
    for n in (0, 1, 5, 4099, 300_001):
        a, b = rng.standard_normal(n).astype(npd), rng.standard_normal(n).astype(npd)
        out = torch.full((1,), 7.0, dtype=torch.float64, device=dev)
        ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
        lo._lib.call("mxlo_dot", ctx.handle, dtype_code(dtype), ptr(ta), ptr(tb), n, ptr(out))
        want = float(a.astype(np.float64) @ b.astype(np.float64))
        assert abs(out.item() - want) <= 1e-12 * max(1.0, np.abs(a.astype(np.float64) * b).sum())
        if n:
            res = torch.full((n,), float("nan"), dtype=dtype, device=dev)
            lo._lib.call("mxlo_householder_apply", ctx.handle, dtype_code(dtype), ptr(res), ptr(ta), ptr(tb), n, 2.0, 0.0,
                         lo._lib.SCALARS_F64 if dtype == torch.float32 else 0, ptr(out))
            c = npd(2) * npd(out.item())
            want = (2.0 * (b - c * a).astype(np.float64)).astype(npd)
            assert np.array_equal(res.cpu().numpy(), want)           # elementwise part is bit-exact given the scalar


def test_tune_keys(lo, dev):
    ctx = lo.get_ctx(dev)
    rng = np.random.default_rng(0)
    n = 3_000_001
    d, v = rng.standard_normal(n), rng.standard_normal(n)
    D = lo.opDiagonal(torch.from_numpy(d).to(dev))
    want = oracle.diag_mul(np.empty(n), d, v, 1.5, 0.0)
    try:
        for key, vals in (("blocks_per_cu", (0, 1, 8)), ("nt_min_bytes", (0, 1 << 40)), ("house_inline_n", (0, 1 << 40)), ("red_blocks_per_cu", (1, 16)),
                          ("house_reverse", (0, 1)), ("house_fused", (0, 1)), ("gemm_tile", (32, 64, 128, -1, 0)),
                          ("extend_tiles_per_block", (1, 8, 0)),
                          ("combine_blocks_per_cu", (0, 4)), ("dots_max_nc", (1, 20))):
            for val in vals:
                ctx.tune(key, val)
                out = torch.empty(n, dtype=torch.float64, device=dev)
                lo.mul(out, D, torch.from_numpy(v).to(dev), 1.5, 0.0)
                assert np.array_equal(out.cpu().numpy(), want)
                h = torch.from_numpy(d / np.linalg.norm(d)).to(dev)
                hv = lo.opHouseholder(h) * torch.from_numpy(v).to(dev)
                ref = oracle.householder_mul(np.empty(n), d / np.linalg.norm(d), v, 1.0, 0.0)
                assert np.linalg.norm(hv.cpu().numpy() - ref) <= 1e-12 * np.linalg.norm(ref)
    finally:
        for key, val in (("blocks_per_cu", 0), ("nt_min_bytes", 256 << 20), ("red_blocks_per_cu", 4), ("house_reverse", 1),
                         ("gemm_tile", 0), ("combine_blocks_per_cu", 0), ("dots_max_nc", 20), ("house_fused", 1),
                         ("extend_tiles_per_block", 0), ("house_inline_n", 1 << 23)):
            ctx.tune(key, val)


@pytest.mark.parametrize("n,world", [(1, 1), (7, 3), (10, 4), (1000, 8), (12345, 7), (5, 8), (2_000_003, 8)])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32, torch.complex128])
def test_shard_stage_pack_unpack_bit_exact(lo, dev, n, world, dtype):
    """mxlo_shard_stage: ShardPlan layout <-> `world` padded slots, both directions, nvalid prefix, bit copies."""
    from linearoperators_jl_amd.device import get_ctx, ptr
    ctx = get_ctx(dev)
    plan = lo.sharded.ShardPlan(n, world)
    pad = -(-n // world)
    rng = np.random.default_rng(n + world)
    np_dt = {torch.float64: np.float64, torch.float32: np.float32, torch.complex128: np.complex128}[dtype]
    full_h = rng.standard_normal(n).astype(np_dt)
    if dtype == torch.complex128:
        full_h = full_h + 1j * rng.standard_normal(n)
    es = full_h.itemsize
    full = torch.from_numpy(full_h).to(dev)
    for nvalid in (-1, n // 3):
        wire = torch.full((world * pad,), 7.0, dtype=dtype, device=dev)        # poison: every slot must be written
        lo._lib.call("mxlo_shard_stage", ctx.handle, es, ptr(wire), ptr(full), n, world, nvalid, lo._lib.SHARD_PACK)
        want = np.zeros(world * pad, dtype=np_dt)
        lim = n if nvalid < 0 else nvalid
        for r in range(world):
            a, b = plan.lo(r), plan.hi(r)
            seg = full_h[a:b].copy()
            seg[max(0, lim - a):] = 0                      # rows >= nvalid count as zeros
            want[r * pad:r * pad + (b - a)] = seg
        assert np.array_equal(wire.cpu().numpy().view(np.uint8), want.view(np.uint8))
    # unpack the nvalid = n packing back
    lo._lib.call("mxlo_shard_stage", ctx.handle, es, ptr(wire), ptr(full), n, world, -1, lo._lib.SHARD_PACK)
    back = torch.zeros(n, dtype=dtype, device=dev)
    lo._lib.call("mxlo_shard_stage", ctx.handle, es, ptr(back), ptr(wire), n, world, -1, lo._lib.SHARD_UNPACK)
    assert torch.equal(back.view(torch.uint8) if dtype != torch.complex128 else torch.view_as_real(back),
                       full.view(torch.uint8) if dtype != torch.complex128 else torch.view_as_real(full))
    assert lo._lib.lib().mxlo_shard_stage(ctx.handle, es, ptr(back), ptr(wire), n, world, -1, 5) == lo._lib.EINVAL
