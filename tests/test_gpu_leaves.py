"""-m gpu: parity of the HIP leaves (through the C ABI, via the host mirror) against the oracle on
seeded inputs, the reference tests' KATs, edge cases and size-independent properties.

Tolerances: elementwise leaves (opDiagonal, opEye, opZeros, scale, restriction/extension) are
BIT-EXACT; leaves with a global reduction (opHouseholder, opOnes) 1e-12 relative L2 in fp64 and
1e-5 in fp32 (fixed-order tree vs the oracle's / BLAS' order)."""
import os
import time

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu

SV = lambda n: np.array([-(-1.0) ** i for i in range(1, n + 1)])
NP = {torch.float64: np.float64, torch.float32: np.float32}
SIZES = [1, 2, 3, 7, 64, 255, 1000, 4097, 100_003, 1_048_577]


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def rel(a, b):
    nb = np.linalg.norm(b.astype(np.float64))
    return np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / (nb if nb else 1.0)


# ---------------------------------------------------------------------------- KATs through the ABI
def test_kat_diag(lo, dev, kat):
    for c in [c for c in kat if c["kind"] == "diag"]:
        d, u = np.array(c["d"]), np.array(c["u"])
        D = lo.opDiagonal(T(d, dev))
        assert np.array_equal((D * T(u, dev)).cpu().numpy(), np.array(c["expect_apply"]))
        assert np.array_equal((D.T * T(u, dev)).cpu().numpy(), np.array(c["expect_apply"]))
        assert np.array_equal((D.H * T(u, dev)).cpu().numpy(), np.array(c["expect_apply"]))
        res = T(np.array(c["res0"]), dev)
        lo.mul(res, D, T(u, dev), c["alpha"], c["beta"])
        assert np.array_equal(res.cpu().numpy(), np.array(c["expect_mul5"]))
    for c in [c for c in kat if c["kind"] == "diag_scalar"]:
        d, u = np.array(c["d"]), np.array(c["u"])
        res = torch.full((u.size,), float("nan"), dtype=torch.float64, device=dev)
        lo.leaves.mulSquareOpDiagonal(res, T(d, dev), T(u, dev), 1.0, 0.0)
        assert np.array_equal(res.cpu().numpy(), np.array(c["expect_apply"]))


def test_kat_diag_rect(lo, dev, kat):
    for c in [c for c in kat if c["kind"] == "diag_rect"]:
        D = lo.opDiagonal(c["nrow"], c["ncol"], T(np.array(c["d"]), dev))
        assert np.array_equal((D * T(np.array(c["u"]), dev)).cpu().numpy(), np.array(c["expect_apply"]))
        assert np.array_equal((D.T * T(np.array(c["w"]), dev)).cpu().numpy(), np.array(c["expect_tapply"]))
        assert np.array_equal((D.H * T(np.array(c["w"]), dev)).cpu().numpy(), np.array(c["expect_tapply"]))


def test_kat_householder(lo, dev, kat):
    for c in [c for c in kat if c["kind"] == "householder"]:
        H = lo.opHouseholder(T(np.array(c["h"]), dev))
        u = T(np.array(c["u"]), dev)
        for op in (H, H.T, H.H):
            assert np.array_equal((op * u).cpu().numpy(), np.array(c["expect_apply"]))


def _idx_arg(lo, spec):
    if "list" in spec:
        return spec["list"]
    if "range" in spec:
        a, b, s = spec["range"]
        return lo.jrange(a, b, s)
    if "scalar" in spec:
        return spec["scalar"]
    return slice(None)


def test_kat_restriction_extension(lo, dev, kat):
    """test_linop.jl:437-461 — all six identities, exact `==`."""
    for c in [c for c in kat if c["kind"] == "restriction"]:
        n = c["n"]
        idx = _idx_arg(lo, c["idx"])
        P = lo.opRestriction(idx, n, device=dev)
        Z = lo.opExtension(idx, n, device=dev)
        v, w, vz = T(np.array(c["v"]), dev), T(np.array(c["expect_w"]), dev), T(np.array(c["expect_vz"]), dev)
        assert torch.equal(P * v, w), c["name"]
        assert torch.equal(P.H * w, vz)
        assert torch.equal(Z * w, vz)
        assert torch.equal(Z.H * v, w)
        assert torch.equal((P * Z) * w, w)
        assert torch.equal((Z * P) * v, vz)


def test_kat_cat_eye_zeros(lo, dev, kat):
    """test_cat.jl:43-49."""
    c = [c for c in kat if c["kind"] == "cat_eye_zeros"][0]
    K = lo.hvcat((2, 2), lo.opEye(2), lo.opZeros(2, 3), lo.opZeros(3, 2), lo.opEye(3))
    v = T(np.array(c["v"]), dev)
    assert torch.equal(K * v, T(np.array(c["expect"]), dev))
    c = [c for c in kat if c["kind"] == "cat_vcat_eye"][0]
    K = lo.vcat(lo.opEye(2), torch.eye(2, dtype=torch.float64, device=dev))
    assert torch.equal(K * T(np.array(c["v"]), dev), T(np.array(c["expect"]), dev))
    # the reference's second block IS sparse (`sparse(1.0I, 2, 2)`): the same identity through the sparse leaf (round 4)
    K = lo.vcat(lo.opEye(2), lo.sparse_csc([1, 2, 3], [1, 2], [1.0, 1.0], 2, 2, index_base=1, device=dev))
    assert torch.equal(K * T(np.array(c["v"]), dev), T(np.array(c["expect"]), dev))
    with pytest.raises(lo.LinearOperatorException):
        lo.vcat(lo.LinearOperatorFromMatrix(torch.ones(5, 5, dtype=torch.float64, device=dev)), lo.opEye(3))


# ---------------------------------------------------------------------------- seeded parity vs oracle
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("n", SIZES)
def test_diag_bit_exact(lo, dev, dtype, n):
    rng = np.random.default_rng(n)
    npd = NP[dtype]
    d, v, r0 = (rng.standard_normal(n).astype(npd) for _ in range(3))
    D = lo.opDiagonal(T(d, dev))
    # all four scalar-type combinations of Julia's mixed-precision rule (src/special-operators.jl:126-129):
    # the α-term is evaluated in α's type, the β-term in β's, the sum in the wider one
    for alpha, beta in ((1.0, 0.0), (2.0 / 3.0, 0.0), (-1.25, 1.0 / 7.0), (0.0, 2.0), (1, 0), (np.float32(0.3), np.float32(0.7)),
                        (np.float32(0.3), 1.0 / 7.0), (2.0 / 3.0, np.float32(0.7)), (np.float32(1.1), 0.0), (1.0 / 3.0, np.float32(0))):
        res = T(r0.copy(), dev)
        if beta == 0:
            res.fill_(float("nan"))                       # beta == 0: res is never read
        lo.mul(res, D, T(v, dev), alpha, beta)
        flags = oracle.scalar_flags(npd, alpha, beta)
        want = oracle.diag_mul(r0.copy(), d, v, float(alpha), float(beta), flags=flags)
        assert np.array_equal(res.cpu().numpy(), want), (alpha, beta)


@pytest.mark.parametrize("n", [1, 255, 4099, 100_003])
def test_mixed_scalar_types_bit_exact_every_elementwise_leaf(lo, dev, n):
    """Float32 data with (Float32, Float64) and (Float64, Float32) scalars through opEye, opZeros, scale,
    the generic axpby of prod3!, BlockDiagonal of diagonals and kron of diagonals — all advertised bit-exact."""
    rng = np.random.default_rng(900 + n)
    d, v, r0 = (rng.standard_normal(n).astype(np.float32) for _ in range(3))
    E = lo.opEye(torch.float32, n, S=lo.Storage(torch.float32, dev))
    Z = lo.opZeros(torch.float32, n, n, S=lo.Storage(torch.float32, dev))
    for alpha, beta in ((np.float32(0.3), 1.0 / 7.0), (2.0 / 3.0, np.float32(0.7)), (np.float32(0.3), np.float32(0.7)), (2.0 / 3.0, 1.0 / 7.0)):
        fl = oracle.scalar_flags(np.float32, alpha, beta)
        res = T(r0.copy(), dev)
        lo.mul(res, E, T(v, dev), alpha, beta)
        assert np.array_equal(res.cpu().numpy(), oracle.eye_mul(r0.copy(), v, float(alpha), float(beta), flags=fl | oracle.TAIL_BETA))
        res = T(r0.copy(), dev)
        lo.mul(res, Z, T(v, dev), alpha, beta)
        assert np.array_equal(res.cpu().numpy(), oracle.zeros_mul(r0.copy(), float(beta), flags=fl))
    if n >= 255:
        m = n // 3
        blocks = [lo.opDiagonal(T(d[:m].copy(), dev)), lo.opDiagonal(T(d[m:2 * m].copy(), dev)), lo.opDiagonal(T(d[2 * m:].copy(), dev))]
        BD = lo.BlockDiagonalOperator(*blocks)
        for alpha, beta in ((np.float32(0.3), 1.0 / 7.0), (2.0 / 3.0, np.float32(0.7))):
            res = T(r0.copy(), dev)
            lo.mul(res, BD, T(v, dev), alpha, beta)
            want = oracle.diag_mul(r0.copy(), d, v, float(alpha), float(beta), flags=oracle.scalar_flags(np.float32, alpha, beta))
            assert np.array_equal(res.cpu().numpy(), want), (alpha, beta)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_diag_misaligned_views(lo, dev, dtype):
    """cat/block-diag hand contiguous views (pointer+offset): every 16-byte phase combination."""
    rng = np.random.default_rng(11)
    npd = NP[dtype]
    n = 5000
    base = [rng.standard_normal(n + 8).astype(npd) for _ in range(3)]
    tb = [T(b, dev) for b in base]
    for od, ov, orr in [(0, 0, 0), (1, 1, 1), (1, 0, 0), (0, 1, 2), (3, 3, 3), (2, 1, 3)]:
        m = n - 3
        d, v, r = base[0][od:od + m], base[1][ov:ov + m], base[2][orr:orr + m].copy()
        res = tb[2].clone()[orr:orr + m]
        lo.leaves.mulSquareOpDiagonal(res, tb[0][od:od + m], tb[1][ov:ov + m], 1.5, -0.5)
        flags = oracle.SCALARS_F64 if dtype == torch.float32 else 0
        want = oracle.diag_mul(r, np.ascontiguousarray(d), np.ascontiguousarray(v), 1.5, -0.5, flags=flags)
        assert np.array_equal(res.cpu().numpy(), want), (od, ov, orr)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("n", SIZES)
def test_householder_parity(lo, dev, dtype, n):
    rng = np.random.default_rng(100 + n)
    npd = NP[dtype]
    h = rng.standard_normal(n)
    h = (h / np.linalg.norm(h)).astype(npd)
    v, r0 = rng.uniform(-1, 1, n).astype(npd), rng.uniform(-1, 1, n).astype(npd)
    H = lo.opHouseholder(T(h, dev))
    tol = 1e-12 if dtype == torch.float64 else 1e-5
    for alpha, beta in ((1.0, 0.0), (2.0, -3.0)):
        res = T(r0.copy(), dev)
        if beta == 0:
            res.fill_(float("nan"))
        lo.mul(res, H, T(v, dev), alpha, beta)
        flags = oracle.SCALARS_F64 if dtype == torch.float32 else 0
        want = oracle.householder_mul(r0.copy(), h, v, alpha, beta, flags=flags)
        assert rel(res.cpu().numpy(), want) <= tol, (n, alpha, beta)
    # deterministic run to run (fixed-order reduction, no float atomics)
    a = H * T(v, dev)
    b = H * T(v, dev)
    assert torch.equal(a, b)


def test_householder_involution_full_size(lo, dev):
    """BASELINE config 2 size (n = 1e8 fp64): H(Hv) == v for ||h|| = 1; parity with the ORACLE (the C restatement of
    mulHouseholder!, src/linalg.jl:77-83) on EVERY element of the output, 1e-12 relative (the one reduction is a fixed-order
    tree on the device, a sequential sum in the oracle); and, given the device's own dot, the elementwise part
    res = v - (2 h'v) h in the reference's rounding order to the bit."""
    from linearoperators_jl_amd import _lib
    from linearoperators_jl_amd.device import dtype_code, get_ctx, ptr
    n = 100_000_000
    g = torch.Generator(device=dev).manual_seed(5)
    h = torch.rand(n, dtype=torch.float64, device=dev, generator=g) - 0.5
    h /= torch.linalg.vector_norm(h)
    v = torch.rand(n, dtype=torch.float64, device=dev, generator=g) * 2 - 1
    H = lo.opHouseholder(h)
    w = H * v
    back = H * w
    err = (torch.linalg.vector_norm(back - v) / torch.linalg.vector_norm(v)).item()
    assert err <= 1e-12, err
    del back
    # the oracle on the same operands, every element
    want = oracle.householder_mul(np.empty(n), h.cpu().numpy(), v.cpu().numpy(), 1.0, 0.0)
    got = w.cpu().numpy()
    assert rel(got, want) <= 1e-12
    assert np.max(np.abs(got - want)) <= 1e-12 * max(1.0, float(np.max(np.abs(want))))
    del want, got
    # the elementwise pass to the bit, given the device's dot (the fixed-order tree mxlo_dot shares with the apply)
    ctx = get_ctx(dev)
    d = torch.zeros(1, dtype=torch.float64, device=dev)
    _lib.call("mxlo_dot", ctx.handle, dtype_code(torch.float64), ptr(h), ptr(v), n, ptr(d))
    c = 2.0 * d                                         # c = 2 * dot(h, v); res = v - c * h (two roundings, no FMA)
    assert torch.equal(w, v - c * h)
    del w
    # opDiagonal at the same size: bit-exact against the same formula evaluated by torch
    D = lo.opDiagonal(h)
    out = torch.empty_like(v)
    lo.mul(out, D, v, 2.0, 0.0)
    assert torch.equal(out, (2.0 * h) * v)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_eye_zeros_ones_scale(lo, dev, dtype):
    rng = np.random.default_rng(3)
    npd = NP[dtype]
    S = lo.Storage(dtype, dev)
    for nrow, ncol in ((7, 7), (9, 4), (4, 9), (1000, 1003)):
        v, r0 = rng.standard_normal(ncol).astype(npd), rng.standard_normal(nrow).astype(npd)
        for alpha, beta in ((1.0, 0.0), (2.5, -0.75)):
            fl = oracle.SCALARS_F64 if dtype == torch.float32 else 0
            E = lo.opEye(dtype, nrow, ncol, S=S)
            res = T(r0.copy(), dev)
            lo.mul(res, E, T(v, dev), alpha, beta)
            want = oracle.eye_mul(r0.copy(), v, alpha, beta, n_min=min(nrow, ncol), flags=fl | oracle.TAIL_BETA)
            assert np.array_equal(res.cpu().numpy(), want)
            Z = lo.opZeros(dtype, nrow, ncol, S=S)
            res = T(r0.copy(), dev)
            lo.mul(res, Z, T(v, dev), alpha, beta)
            assert np.array_equal(res.cpu().numpy(), oracle.zeros_mul(r0.copy(), beta, flags=fl))
            O = lo.opOnes(dtype, nrow, ncol, S=S)
            res = T(r0.copy(), dev)
            lo.mul(res, O, T(v, dev), alpha, beta)
            want = oracle.ones_mul(r0.copy(), v, alpha, beta, flags=fl)
            assert rel(res.cpu().numpy(), want) <= (1e-12 if dtype == torch.float64 else 1e-5)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32, torch.int64])
def test_restriction_extension_bit_exact(lo, dev, dtype):
    rng = np.random.default_rng(8)
    n = 200_003
    if dtype == torch.int64:
        v = rng.integers(-2**62, 2**62, n)
    else:
        v = rng.standard_normal(n).astype(NP[dtype])
        v[::97] = np.nan                                   # payload bits must survive
    vt = T(v, dev)
    specs = [rng.integers(1, n + 1, 50_000), np.arange(1, n + 1), np.array([n]), np.array([], dtype=np.int64),
             rng.permutation(n)[:1000] + 1]
    for idx in specs:
        P = lo.opRestriction(idx, n, device=dev)
        out = torch.empty(len(idx), dtype=dtype, device=dev)
        lo.mul(out, P, vt)
        assert np.array_equal(out.cpu().numpy().view(np.uint8), v[np.asarray(idx, dtype=np.int64) - 1].view(np.uint8))
        u = v[:len(idx)].copy()
        back = torch.full((n,), 7, dtype=dtype, device=dev)
        lo.mul(back, P.H, T(u, dev))
        want = oracle.extend(np.empty(n, v.dtype), u, np.asarray(idx, dtype=np.int64))
        assert np.array_equal(back.cpu().numpy().view(np.uint8), want.view(np.uint8))
    for (a, b, s) in ((3, 6, 1), (1, n, 2), (n, 1, -3), (5, 4, 1), (17, 17, 1)):
        r = lo.jrange(a, b, s)
        idx = r.to_numpy()
        P = lo.opRestriction(r, n, device=dev)
        out = torch.empty(len(r), dtype=dtype, device=dev)
        lo.mul(out, P, vt)
        assert np.array_equal(out.cpu().numpy().view(np.uint8), v[idx - 1].view(np.uint8)), (a, b, s)
        u = v[:len(r)].copy()
        back = torch.full((n,), 7, dtype=dtype, device=dev)
        lo.mul(back, P.H, T(u, dev))
        want = oracle.extend(np.empty(n, v.dtype), u, idx)
        assert np.array_equal(back.cpu().numpy().view(np.uint8), want.view(np.uint8)), (a, b, s)
    with pytest.raises(lo.LinearOperatorException):
        lo.opRestriction([0, 1], n, device=dev)
    with pytest.raises(lo.LinearOperatorException):
        lo.opRestriction([n + 1], n, device=dev)


def test_index_plan_form_matches_the_index_list_form(lo, dev, monkeypatch):
    """Round 5: a dense strictly increasing index set is applied as bit mask + ranks (no index list
    read). Through the operator API, against the index-list kernels on the same inputs (PLAN_MIN_DENSITY = 0) and numpy:
    Float64 / Float32 / ComplexF64, increasing I (both applies use the plan), a permuted I with duplicates (extension uses
    the plan + pos, restriction keeps the list), NaN payloads, > 1/32 and < 1/32 densities; and the refusals of the ABI."""
    import ctypes as C
    from linearoperators_jl_amd import leaves
    from linearoperators_jl_amd.device import get_ctx
    rng = np.random.default_rng(21)
    n = 300_007
    for dtype in (torch.float64, torch.float32, torch.complex128):
        v = rng.standard_normal(n) if dtype != torch.complex128 else rng.standard_normal(n) + 1j * rng.standard_normal(n)
        v = v.astype({torch.float64: np.float64, torch.float32: np.float32, torch.complex128: np.complex128}[dtype])
        v[::89] = np.nan
        vt = T(v, dev)
        for idx in (np.flatnonzero(rng.random(n) < 0.5) + 1, np.flatnonzero(rng.random(n) < 0.04) + 1,
                    np.flatnonzero(rng.random(n) < 0.01) + 1, rng.integers(1, n + 1, 150_000), np.arange(1, n + 1)):
            outs = []
            for plan_on in (1, 0):
                monkeypatch.setattr(leaves, "PLAN_MIN_DENSITY", plan_on)
                P = lo.opRestriction(idx, n, device=dev)
                out = torch.empty(len(idx), dtype=dtype, device=dev)
                lo.mul(out, P, vt)
                back = torch.full((n,), 7, dtype=dtype, device=dev)
                lo.mul(back, P.H, vt[:len(idx)])
                outs.append((out.cpu().numpy(), back.cpu().numpy()))
            b = lambda a: np.ascontiguousarray(a).view(np.uint8)
            assert np.array_equal(b(outs[0][0]), b(outs[1][0])) and np.array_equal(b(outs[0][1]), b(outs[1][1]))
            assert np.array_equal(b(outs[0][0]), b(v[np.asarray(idx) - 1]))
            want = oracle.extend(np.empty(n, v.dtype), v[:len(idx)].copy(), np.asarray(idx, dtype=np.int64)) if dtype != torch.complex128 else None
            if want is not None:
                assert np.array_equal(b(outs[0][1]), b(want))
    ctx = get_ctx(dev)
    plan = C.c_void_p()
    bad = np.array([1, 5, 5, 9], dtype=np.int64)
    with pytest.raises(Exception, match="strictly increasing"):
        lo._lib.call("mxlo_index_plan_create", ctx.handle, bad.ctypes.data, 4, 10, C.byref(plan))
    good = np.array([1, 5, 9], dtype=np.int64)
    with pytest.raises(Exception, match="strictly increasing"):
        lo._lib.call("mxlo_index_plan_create", ctx.handle, good.ctypes.data, 3, 8, C.byref(plan))       # 9 > n
    lo._lib.call("mxlo_index_plan_create", ctx.handle, good.ctypes.data, 3, 10, C.byref(plan))
    x = torch.zeros(11, dtype=torch.float64, device=dev)
    with pytest.raises(Exception, match="built for 10"):
        lo._lib.call("mxlo_gather_plan", ctx.handle, 8, x.data_ptr(), x.data_ptr(), 11, plan)
    lo._lib.call("mxlo_index_plan_destroy", plan)


def test_extension_duplicates_last_write_wins(lo, dev):
    idx = np.array([2, 5, 2, 3, 5, 5], dtype=np.int64)
    u = np.arange(10.0, 70.0, 10.0)
    Z = lo.opExtension(idx, 6, device=dev)
    got = (Z * T(u, dev)).cpu().numpy()
    assert np.array_equal(got, oracle.extend(np.empty(6), u, idx))


def test_shape_mismatch_and_errors(lo, dev):
    D = lo.opDiagonal(torch.ones(5, dtype=torch.float64, device=dev))
    with pytest.raises(lo.LinearOperatorException, match="shape mismatch"):
        D * torch.ones(6, dtype=torch.float64, device=dev)
    with pytest.raises(lo.LinearOperatorException):
        D.size(3)
    with pytest.raises(RuntimeError):                      # CPU operands: no fallback
        lo.opDiagonal(torch.ones(5, dtype=torch.float64))


def test_counters_and_prod3(lo, dev):
    """test_linop.jl:634-716 counters; :768-817 3-arg closures with lazily allocated Mv."""
    S = lo.Storage(torch.float64, dev)
    d = torch.arange(1.0, 6.0, dtype=torch.float64, device=dev)
    calls = []

    def p3(res, v):
        calls.append(1)
        res.copy_(d * v)

    op = lo.LinearOperator(torch.float64, 5, 5, True, True, p3, None, None, S=S)
    assert not lo.has_args5(op) and not lo.isallocated5(op)
    v = torch.ones(5, dtype=torch.float64, device=dev)
    res = torch.empty(5, dtype=torch.float64, device=dev)
    lo.mul(res, op, v)
    assert torch.equal(res, d) and not lo.isallocated5(op)
    lo.mul(res, op, v, 2.0, 0.0)
    assert torch.equal(res, 2 * d) and not lo.isallocated5(op)
    lo.mul(res, op, v, 2.0, 3.0)
    assert torch.equal(res, 2 * d + 3 * (2 * d)) and lo.isallocated5(op)
    assert lo.nprod(op) == 3
    lo.mul(res, op.T, v)
    lo.mul(res, op.H, v)
    assert lo.nprod(op) == 5 and lo.ntprod(op) == 0 and lo.nctprod(op) == 0   # symmetric+hermitian -> prod!
    lo.reset(op)
    assert lo.nprod(op) == 0


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_empty_operators(lo, dev, dtype):
    """Zero-length operands (SURVEY §8a "empty and ragged inputs"): every leaf must accept n = 0 and do nothing."""
    S = lo.Storage(dtype, dev)
    e = torch.empty(0, dtype=dtype, device=dev)
    ops = [lo.opDiagonal(e), lo.opHouseholder(e), lo.opEye(dtype, 0, S=S), lo.opZeros(dtype, 0, 0, S=S),
           lo.opOnes(dtype, 0, 0, S=S), lo.opHermitian(e, torch.empty(0, 0, dtype=dtype, device=dev)),
           lo.LinearOperatorFromMatrix(torch.empty(0, 0, dtype=dtype, device=dev)),
           lo.InverseLBFGSOperator(dtype, 0, mem=3, device=dev), lo.LBFGSOperator(dtype, 0, mem=3, device=dev),
           lo.LSR1Operator(dtype, 0, mem=3, device=dev)]
    for op in ops:
        assert op.shape == (0, 0)
        for o in (op, op.T, op.H):
            out = o * e
            assert out.numel() == 0
        res = torch.empty(0, dtype=dtype, device=dev)
        lo.mul(res, op, e, 2.0, -3.0)
        assert lo.Matrix(op).shape == (0, 0)
    # rectangular cases with one empty side: m x 0 and 0 x n
    v5 = torch.arange(1, 6, dtype=dtype, device=dev)
    Z = lo.opZeros(dtype, 5, 0, S=S)
    r = torch.full((5,), 7.0, dtype=dtype, device=dev)
    lo.mul(r, Z, e, 2.0, 0.0)
    assert torch.equal(r, torch.zeros_like(r))
    R = lo.opRestriction([], 5, S=S)                                    # empty index set
    assert (R * v5).numel() == 0
    assert torch.equal(R.T * e, torch.zeros(5, dtype=dtype, device=dev))   # res .= 0; res[[]] = []
    M = lo.LinearOperatorFromMatrix(torch.empty(5, 0, dtype=dtype, device=dev))
    r = torch.full((5,), 7.0, dtype=dtype, device=dev)
    lo.mul(r, M, e, 2.0, -3.0)                                          # empty sum: res = β·res
    assert torch.equal(r, torch.full_like(r, -21.0))
    assert (M.T * v5).numel() == 0
    K = lo.kron(torch.empty(0, 0, dtype=dtype, device=dev), torch.ones(3, 3, dtype=dtype, device=dev))
    assert K.shape == (0, 0) and (K * e).numel() == 0
    Bd = lo.BlockDiagonalOperator(lo.opDiagonal(e), lo.opDiagonal(v5))
    assert torch.equal(Bd * v5, v5 * v5)


@pytest.mark.parametrize("es", [4, 8, 16])
def test_sorted_extension_and_range_kernels_bit_exact_edge_shapes(lo, dev, es):
    """mxlo_scatter_zero_sorted (segment-owner tiles), mxlo_scatter_zero (memset + scatter), mxlo_scatter_zero_range and
    mxlo_gather_range (division-free 16-byte lanes) against `res .= 0; res[I] = u` / `v[I]` on numpy: dense, sparse and
    clustered plans, tile-boundary sizes, unaligned res, pos maps, negative and wide steps, all three element sizes."""
    from linearoperators_jl_amd.device import get_ctx, ptr
    ctx = get_ctx(dev)
    rng = np.random.default_rng(100 + es)
    words = es // 4
    tile = 16384 // es

    def rand_elems(n):
        return rng.integers(1, 2**32, (n, words), dtype=np.uint64).astype(np.uint32)     # never all-zero elements

    def dev_elems(a, off=0):
        """upload with the first element `off` elements past a 16-byte boundary"""
        buf = torch.zeros((a.shape[0] + 4) * words, dtype=torch.int32, device=dev)
        view = buf[off * words:(off + a.shape[0]) * words]
        view.copy_(torch.from_numpy(a.view(np.int32).reshape(-1)).to(dev))
        return buf, view

    sizes = [1, 3, tile - 1, tile, tile + 1, 3 * tile + 5, 200_003]
    for nres in sizes:
        plans = []
        dense = np.flatnonzero(rng.random(nres) < 0.5) + 1
        plans.append(dense)
        plans.append(np.arange(1, nres + 1))                                             # every slot
        plans.append(np.array([], dtype=np.int64))
        plans.append(np.array([nres], dtype=np.int64))
        plans.append(np.array([1], dtype=np.int64))
        if nres > 10:
            plans.append(np.unique(rng.integers(1, nres + 1, max(1, nres // 300))))      # sparse
            plans.append(np.arange(1, nres // 2))                                        # one cluster + a giant gap
        for off in (0, 1):
            if es == 16 and off:
                continue
            for idx in plans:
                nidx = idx.size
                u = rand_elems(max(nidx, 1) + 3)
                use_pos = nidx > 1 and rng.random() < 0.5
                pos = rng.permutation(u.shape[0])[:nidx].astype(np.int64) if use_pos else None
                want = np.zeros((nres, words), dtype=np.uint32)
                if nidx:
                    want[idx - 1] = u[pos] if use_pos else u[:nidx]
                _, ud = dev_elems(u)
                idx_d = torch.from_numpy(idx.astype(np.int64)).to(dev)
                pos_d = torch.from_numpy(pos).to(dev) if use_pos else None
                for fn in ("mxlo_scatter_zero_sorted", "mxlo_scatter_zero"):
                    rbuf, rd = dev_elems(np.full((nres, words), 0xDEADBEEF, dtype=np.uint32), off)
                    lo._lib.call(fn, ctx.handle, es, ptr(rd), nres, ptr(ud), ptr(idx_d), ptr(pos_d), nidx)
                    got = rd.cpu().numpy().view(np.uint32).reshape(nres, words)
                    assert np.array_equal(got, want), (fn, nres, nidx, off, use_pos)
                    guard = rbuf.cpu().numpy()
                    assert not guard[:off * words].any() and not guard[(off + nres) * words:].any(), "wrote outside res"
                # round 5: the same index set as bit mask + ranks (mxlo_index_plan): extension, and the restriction back
                import ctypes as C
                plan = C.c_void_p()
                ih = np.ascontiguousarray(idx, dtype=np.int64)
                lo._lib.call("mxlo_index_plan_create", ctx.handle, ih.ctypes.data, nidx, nres, C.byref(plan))
                try:
                    rbuf, rd = dev_elems(np.full((nres, words), 0xDEADBEEF, dtype=np.uint32), off)
                    lo._lib.call("mxlo_scatter_zero_plan", ctx.handle, es, ptr(rd), nres, ptr(ud), ptr(pos_d), plan)
                    got = rd.cpu().numpy().view(np.uint32).reshape(nres, words)
                    assert np.array_equal(got, want), ("plan extension", nres, nidx, off, use_pos)
                    guard = rbuf.cpu().numpy()
                    assert not guard[:off * words].any() and not guard[(off + nres) * words:].any(), "wrote outside res"
                    vsrc = rand_elems(nres)
                    _, vd2 = dev_elems(vsrc, off)                                       # v at both 16-byte phases
                    gbuf, gd = dev_elems(np.full((max(nidx, 1), words), 0xDEADBEEF, dtype=np.uint32), 1 if es < 16 else 0)
                    lo._lib.call("mxlo_gather_plan", ctx.handle, es, ptr(gd), ptr(vd2), nres, plan)
                    gg = gd.cpu().numpy().view(np.uint32).reshape(-1, words)
                    assert np.array_equal(gg[:nidx], vsrc[idx - 1]), ("plan restriction", nres, nidx, off)
                    assert (gg[nidx:] == 0xDEADBEEF).all(), "wrote past the last selected element"
                finally:
                    lo._lib.call("mxlo_index_plan_destroy", plan)
    # ranges
    n = 100_003
    v = rand_elems(n)
    for off in (0, 1):
        if es == 16 and off:
            continue
        _, vd = dev_elems(v)
        for (a, b, s) in ((1, n, 1), (1, n, 2), (2, n, 2), (n, 1, -1), (n, 2, -2), (7, n - 5, 3), (n - 1, 4, -5), (1, n, 4),
                          (3, n, 64), (1, n, n - 1), (10, 10, 1), (5, 4, 1), (n, n, -7), (1, n, 1000), (2, 2 + 13 * 17, 17)):
            r = lo.jrange(a, b, s)
            idx = r.to_numpy()
            ln = len(r)
            rbuf, rd = dev_elems(np.full((max(ln, 1), words), 0xDEADBEEF, dtype=np.uint32), off)
            lo._lib.call("mxlo_gather_range", ctx.handle, es, ptr(rd), ptr(vd), n, a, s, ln)
            got = rd.cpu().numpy().view(np.uint32).reshape(-1, words)[:ln]
            assert np.array_equal(got, v[idx - 1]), ("gather", a, b, s, off)
            u = rand_elems(max(ln, 1))
            _, ud = dev_elems(u)
            rbuf, rd = dev_elems(np.full((n, words), 0xDEADBEEF, dtype=np.uint32), off)
            lo._lib.call("mxlo_scatter_zero_range", ctx.handle, es, ptr(rd), n, ptr(ud), a, s, ln)
            want = np.zeros((n, words), dtype=np.uint32)
            want[idx - 1] = u[:ln]
            got = rd.cpu().numpy().view(np.uint32).reshape(n, words)
            assert np.array_equal(got, want), ("extend", a, b, s, off)
            guard = rbuf.cpu().numpy()
            assert not guard[:off * words].any() and not guard[(off + n) * words:].any()


def test_single_launch_householder_matches_two_pass_and_oracle(lo, dev):
    """n <= 2^22 doubles (round 6; rounds 3-5: 2^21) runs ONE kernel (register-resident slices + slot exchange between workgroups). Checked against the
    oracle (1e-12 / 1e-5) and against the two-launch path (tune house_fused = 0): alternating grid sizes (slot re-arming,
    epoch flips), unaligned views, beta != 0, mixed-precision scalars, NaN partials, graph replay."""
    ctx = lo.get_ctx(dev)
    rng = np.random.default_rng(77)
    # (round 6: up to TWO co-resident workgroups per CU — 257 ... 512 workgroups, n up to 2^22 doubles: the last four sizes)
    sizes = [1, 2, 3, 255, 1024, 1025, 65_536, 4097, 1 << 20, 1000, (1 << 20) + 1, 300_001, 17, (1 << 21) + 7, 3_000_001, 1 << 22, 1 << 21]
    for dtype, tol in ((torch.float64, 1e-12), (torch.float32, 2e-5)):
        for n in sizes:
            for off in (0, 1):
                h = rng.standard_normal(n + off).astype(NP[dtype])[off:]
                h /= np.linalg.norm(h.astype(np.float64)) or 1.0
                v = rng.uniform(-1, 1, n + off).astype(NP[dtype])[off:]
                r0 = rng.uniform(-1, 1, n + off).astype(NP[dtype])[off:]
                hb, vb, rb = (torch.from_numpy(np.concatenate([np.zeros(off, x.dtype), x])).to(dev) for x in (h, v, r0))
                ht, vt = hb[off:], vb[off:]
                H = lo.opHouseholder(ht)
                for alpha, beta in ((1.0, 0.0), (2.0, -3.0), (np.float32(0.5), 0.25)):
                    res = rb.clone()[off:]
                    lo.mul(res, H, vt, alpha, beta)
                    fl = oracle.scalar_flags(NP[dtype], alpha, beta)
                    want = oracle.householder_mul(r0.copy(), h, v, float(alpha), float(beta), flags=fl)
                    assert rel(res.cpu().numpy(), want) <= tol, (dtype, n, off, alpha, beta)
                    ctx.tune("house_fused", 0)
                    try:
                        res2 = rb.clone()[off:]
                        lo.mul(res2, H, vt, alpha, beta)
                    finally:
                        ctx.tune("house_fused", 1)
                    assert rel(res.cpu().numpy(), res2.cpu().numpy()) <= tol
    # the 257 ... 512-workgroup range really is ONE launch, and one workgroup per CU (tune house_fused_per_cu = 1) sends it to two passes
    def launches():
        import ctypes as C
        a = (C.c_int64 * 12)()
        lo._lib.call("mxlo_debug_counters", a)
        return a[10]
    n = 3_000_001
    h = torch.rand(n, dtype=torch.float64, device=dev)
    v, res = torch.rand(n, dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.float64, device=dev)
    H = lo.opHouseholder(h)
    lo.mul(res, H, v, 1.0, 0.0)
    l0 = launches()
    lo.mul(res, H, v, 1.0, 0.0)
    assert launches() - l0 == 1
    ctx.tune("house_fused_per_cu", 1)
    try:
        l0 = launches()
        lo.mul(res, H, v, 1.0, 0.0)
        assert launches() - l0 >= 2
    finally:
        ctx.tune("house_fused_per_cu", 2)
    # a NaN in v poisons the dot: every element becomes NaN (and the exchange does not hang on a NaN partial)
    n = 70_000
    h = torch.full((n,), 1.0 / np.sqrt(n), dtype=torch.float64, device=dev)
    v = torch.ones(n, dtype=torch.float64, device=dev)
    v[12345] = float("nan")
    out = lo.opHouseholder(h) * v
    assert bool(torch.isnan(out).all())
    v[12345] = 1.0
    out = lo.opHouseholder(h) * v                                       # slots are clean again
    assert rel(out.cpu().numpy(), oracle.householder_mul(np.empty(n), h.cpu().numpy(), v.cpu().numpy(), 1.0, 0.0)) <= 1e-12
    # graph replay: the epoch lives in device memory, so one captured launch can be replayed any number of times
    res = torch.empty(n, dtype=torch.float64, device=dev)
    g = lo.capture_mul(res, lo.opHouseholder(h), v, 1.0, 0.0)
    for k in range(5):
        v.mul_(1.0 + 0.1 * k)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        want = oracle.householder_mul(np.empty(n), h.cpu().numpy(), v.cpu().numpy(), 1.0, 0.0)
        assert rel(res.cpu().numpy(), want) <= 1e-12, k


def test_single_launch_householder_exchange_stress(lo, dev):
    """30,000 back-to-back single-launch applies with the grid size changing from call to call (1 ... 256 workgroups),
    interleaved with the two-launch path, graph replays and a long streaming kernel: the slot exchange must neither hang
    nor leak a stale partial (tools/stress_fused_householder.py runs the same loop 300,000 times)."""
    import subprocess
    import sys as _sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, STRESS_CALLS="30000")
    out = subprocess.run([_sys.executable, os.path.join(root, "tools", "stress_fused_householder.py")], env=env,
                         capture_output=True, text=True, timeout=240)
    assert out.returncode == 0 and "no hang" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_single_launch_householder_two_streams_one_ctx(lo, dev):
    """VERDICT r2 (#8): all fused launches of a ctx share one set of exchange slots + one epoch word, which is only
    correct while at most one of them is in flight. `mxlo_ctx_set_stream` keeps that true for callers that alternate
    streams: a stream change records an event on the old stream and makes the new one wait for it (include/mxlo.h), so
    two applies of one ctx are always ordered even when they are issued on two streams. 4,000 alternating single-launch
    applies on two non-blocking torch streams (different operators, different grid sizes), then both results against
    the oracle — a slot-set collision would hang or leak a partial of the other operator into the dot."""
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    rng = np.random.default_rng(91)
    ops = []
    for n in (1 << 16, 40_000):
        h = rng.standard_normal(n)
        h /= np.linalg.norm(h)
        v = rng.uniform(-1, 1, n)
        ops.append((lo.opHouseholder(torch.from_numpy(h).to(dev)), torch.from_numpy(v).to(dev),
                    torch.empty(n, dtype=torch.float64, device=dev), h, v))
    torch.cuda.synchronize()
    for it in range(2000):
        for (H, v, res, _, _), st in zip(ops, (s1, s2)):
            with torch.cuda.stream(st):
                lo.mul(res, H, v, 1.0 + it, 0.0)
    torch.cuda.synchronize()
    for H, v, res, h_np, v_np in ops:
        want = oracle.householder_mul(np.empty(h_np.size), h_np, v_np, 2000.0, 0.0)
        assert rel(res.cpu().numpy(), want) <= 1e-12


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_householder_update_pass_summing_the_partials_is_bit_identical(lo, dev, dtype):
    """Mid sizes (above the single-launch limit, up to `house_inline_n`): the update pass adds up the dots pass's
    per-workgroup partial sums itself, in the finalize kernel's order — one launch fewer, and bit for bit the result of
    the three-launch form (`house_inline_n` = 0), for aligned and ragged n, β = 0 and β ≠ 0, a misaligned view."""
    from linearoperators_jl_amd.device import get_ctx
    ctx = get_ctx(dev)
    gen = torch.Generator(device=dev).manual_seed(5)
    try:
        for n in (1 << 20 | 1, 3_000_001, 1 << 22, (1 << 23) - 3):
            hb = torch.rand(n + 1, dtype=dtype, device=dev, generator=gen) - 0.5
            hb /= hb.norm()
            vb = torch.rand(n + 1, dtype=dtype, device=dev, generator=gen) * 2 - 1
            r0 = torch.rand(n, dtype=dtype, device=dev, generator=gen)
            for off in (0, 1):
                h, v = hb[off:off + n], vb[off:off + n]
                H = lo.opHouseholder(h)
                for a, b in ((1.0, 0.0), (2.0, -3.0)):
                    got = {}
                    for inline in (1 << 23, 0):
                        ctx.tune("house_inline_n", inline)
                        res = r0.clone()
                        lo.mul(res, H, v, a, b)
                        got[inline] = res
                    assert torch.equal(got[1 << 23], got[0]), (n, off, a, b)
    finally:
        ctx.tune("house_inline_n", 1 << 23)


def test_single_launch_householder_timeout_is_an_error_not_a_hang(lo, dev):
    """the workgroups of the single-launch apply wait for each other. If one of them never publishes
    (not co-resident: GPU shared, CU masking; here: the `fused_debug_drop` test hook), the wait must END — NaN result,
    ctx fault word raised — and the next call must say so, repair the exchange state and leave the ctx usable."""
    from linearoperators_jl_amd.device import get_ctx
    ctx = get_ctx(dev)
    rng = np.random.default_rng(17)
    n = 1 << 16
    h = rng.standard_normal(n)
    h /= np.linalg.norm(h)
    v = rng.uniform(-1, 1, n)
    H = lo.opHouseholder(torch.from_numpy(h).to(dev))
    dv = torch.from_numpy(v).to(dev)
    res = torch.zeros(n, dtype=torch.float64, device=dev)
    want = oracle.householder_mul(np.empty(n), h, v, 1.0, 0.0)
    try:
        lo.mul(res, H, dv, 1.0, 0.0)
        torch.cuda.synchronize()
        assert rel(res.cpu().numpy(), want) <= 1e-12
        ctx.tune("fused_timeout_ms", 20)
        ctx.tune("fused_debug_drop", 1)
        t0 = time.perf_counter()
        lo.mul(res, H, dv, 1.0, 0.0)                 # the launch itself succeeds; its workgroups give up after ~20 ms
        torch.cuda.synchronize()
        assert time.perf_counter() - t0 < 5.0
        assert bool(torch.isnan(res).all()), "a timed-out single-launch apply must not leave plausible numbers behind"
        ctx.tune("fused_debug_drop", -1)
        with pytest.raises(Exception, match="timed out"):
            lo.mul(res, H, dv, 1.0, 0.0)             # reported (and repaired) at the next single-launch apply
        lo.mul(res, H, dv, 1.0, 0.0)                 # single-launch forms are off now: two passes
        torch.cuda.synchronize()
        assert rel(res.cpu().numpy(), want) <= 1e-12
        ctx.tune("house_fused", 1)                   # the exchange slots were re-armed: the single launch works again
        for _ in range(5):
            res.zero_()
            lo.mul(res, H, dv, 1.0, 0.0)
        torch.cuda.synchronize()
        assert rel(res.cpu().numpy(), want) <= 1e-12
        # the same fault is also reported by mxlo_ctx_sync when no further apply follows; and the wait lasts what the
        # tune key says (the device's constant-rate clock, hipDeviceAttributeWallClockRate): 300 ms here
        ctx.tune("fused_debug_drop", 0)
        ctx.tune("fused_timeout_ms", 300)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lo.mul(res, H, dv, 1.0, 0.0)
        with pytest.raises(Exception, match="timed out"):
            ctx.sync()
        waited = time.perf_counter() - t0
        assert 0.25 <= waited <= 1.5, waited
        ctx.tune("fused_timeout_ms", 20)
        ctx.tune("fused_debug_drop", -1)
        ctx.tune("house_fused", 1)
        lo.mul(res, H, dv, 1.0, 0.0)
        ctx.sync()
        assert rel(res.cpu().numpy(), want) <= 1e-12
    finally:
        ctx.tune("fused_debug_drop", -1)
        ctx.tune("fused_timeout_ms", 2000)
        ctx.tune("house_fused", 1)
        ctx.tune("qn_fused_small", 1)
        ctx.tune("qn_persist", 1)
        ctx.tune("herm_single", 1)
        ctx.tune("kron_fuse", 1)


def test_graph_replay_on_its_capture_stream_is_ordered_with_the_ctx_stream(lo, dev):
    """ADVICE r3 #1 (second half): a captured single-launch apply replays on the stream it was captured from, while the
    ctx may meanwhile issue direct single-launch applies on another stream — both use the ctx's one set of exchange
    slots. `mxlo_graph_launch` orders the two streams itself (event both ways), so the caller need not."""
    from linearoperators_jl_amd.graph import capture_mul
    rng = np.random.default_rng(23)
    ops = []
    for n in (1 << 15, 50_000):
        h = rng.standard_normal(n)
        h /= np.linalg.norm(h)
        v = rng.uniform(-1, 1, n)
        ops.append((lo.opHouseholder(torch.from_numpy(h).to(dev)), torch.from_numpy(v).to(dev),
                    torch.empty(n, dtype=torch.float64, device=dev), h, v))
    (H0, v0, r0, h0, x0), (H1, v1, r1, h1, x1) = ops
    g = capture_mul(r0, H0, v0, 3.0, 0.0)
    torch.cuda.synchronize()
    for it in range(2000):
        g.replay(sync_streams=False)                 # on the capture stream
        lo.mul(r1, H1, v1, 1.0 + it, 0.0)            # on torch's current stream
    torch.cuda.synchronize()
    assert rel(r0.cpu().numpy(), oracle.householder_mul(np.empty(h0.size), h0, x0, 3.0, 0.0)) <= 1e-12
    assert rel(r1.cpu().numpy(), oracle.householder_mul(np.empty(h1.size), h1, x1, 2000.0, 0.0)) <= 1e-12
