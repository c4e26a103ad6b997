"""CPU-only: the C-ABI library loads and exports every symbol include/mxlo.h declares, the ctypes
prototypes cover the header, and the host-side logic (index ranges, storage promotion, shard plans,
argument-count dispatch) behaves like the reference. No compute call is made (no GPU here)."""
import ctypes
import os

import numpy as np
import pytest
import torch


def test_library_exports_every_header_symbol(lo):
    assert os.path.exists(lo._lib.LIB_PATH), "run __graft_entry__.build() first"
    L = ctypes.CDLL(lo._lib.LIB_PATH)
    syms = lo._lib.header_symbols()
    assert len(syms) >= 50
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing
    L.mxlo_version.restype = ctypes.c_char_p
    assert b"gfx950" in L.mxlo_version()
    L.mxlo_status_string.restype = ctypes.c_char_p
    assert L.mxlo_status_string(2) == b"shape mismatch"


def test_rccl_transport_library_exports_its_header(lo):
    """include/mxlo_rccl.h <-> libmxlo_rccl.so (loading it pulls librccl; no collective is started here)."""
    assert os.path.exists(lo._lib.RCCL_LIB_PATH), "run __graft_entry__.build() first"
    import torch  # noqa: F401  (maps torch's librccl.so.1 so the SONAME resolves the same way as at run time)
    R = ctypes.CDLL(lo._lib.RCCL_LIB_PATH)
    syms = lo._lib.header_symbols(lo._lib.RCCL_HEADER)
    assert {"mxlo_rccl_unique_id", "mxlo_rccl_comm_create", "mxlo_rccl_comm_destroy", "mxlo_rccl_allreduce_hook",
            "mxlo_rccl_last_error", "mxlo_shard_ctx_create", "mxlo_shard_ctx_destroy", "mxlo_shard_ctx_get",
            "mxlo_householder_mul_sharded", "mxlo_diag_mul_sharded", "mxlo_qn_create_sharded", "mxlo_qn_push_sharded",
            "mxlo_qn_mul_sharded", "mxlo_qn_mul_shifted_sharded", "mxlo_qn_solve_shifted_sharded",
            "mxlo_qn_diag_sharded", "mxlo_qn_reset_sharded", "mxlo_qn_get_scalars_sharded"} <= set(syms)
    assert len(syms) == 40
    assert not [s for s in syms if not hasattr(R, s)]


def test_ctypes_prototypes_cover_header(lo):
    declared = set(lo._lib.header_symbols()) - {"mxlo_version", "mxlo_status_string", "mxlo_last_error"}
    assert declared == set(lo._lib._PROTOS), declared ^ set(lo._lib._PROTOS)


def test_every_tune_key_is_documented_in_the_header():
    """`mxlo_ctx_tune` keys are part of the boundary (tools and tests set them through the ABI): every key the library
    accepts must be named in include/mxlo.h."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "linearoperators.jl_amd", "csrc", "api_ctx.hip")).read()
    hdr = open(os.path.join(root, "include", "mxlo.h")).read()
    keys = set(re.findall(r'strcmp\(key, "([a-z_0-9]+)"\)', src))
    assert len(keys) >= 15
    missing = sorted(k for k in keys if f'"{k}"' not in hdr)
    assert not missing, missing


def _gfx950_code_objects(lo, td):
    """the gfx950 code objects of libmxlo.so (clang offload bundles inside the HIP fat binary), written into `td`"""
    import os
    import re
    import struct
    so = os.path.join(os.path.dirname(lo._lib.__file__), "csrc", "libmxlo.so")
    blob = open(so, "rb").read()
    out = []
    for bi, m in enumerate(re.finditer(re.escape(b"__CLANG_OFFLOAD_BUNDLE__"), blob)):
        p = m.start()
        (n,) = struct.unpack_from("<Q", blob, p + 24)
        off = p + 32
        for e in range(n):
            o, size, tl = struct.unpack_from("<QQQ", blob, off)
            off += 24
            triple = blob[off:off + tl].decode()
            off += tl
            if "gfx950" not in triple or size == 0:
                continue
            f = os.path.join(td, f"co_{bi}_{e}.co")
            with open(f, "wb") as fh:
                fh.write(blob[p + o:p + o + size])
            out.append(f)
    return out


def test_launch_bound_push_pass_keeps_its_loads_in_flight(lo):
    """Static check on the built code objects (no GPU needed). Round 4 found the push pass at launch-bound sizes waiting
    for EVERY load before issuing the next (`s_waitcnt vmcnt(0)` in front of each guarded load: 14 memory round trips for
    5 columns, 44 for 20 — 6 .. 12 us for a few KB of data). Its FAST instantiation must issue the
    (NC + 2) * UNROLL loads of a chunk back to back; the instantiation the HBM-bound sizes take must stay within the
    register budget that gives it its occupancy (the branch-free form costs it 6 %)."""
    import os
    import re
    import subprocess
    import tempfile
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        pytest.skip("llvm-objdump / llvm-readelf not available")
    # push_pass_kernel<double, 5, 2, NT=false, STORE=true, FAST>(PushPassArgs<double, 5>, double*)
    fast = "_ZN4mxlo16push_pass_kernelIdLi5ELi2ELb0ELb1ELb1EEEvNS_12PushPassArgsIT_XT0_EEEPd"
    slow = "_ZN4mxlo16push_pass_kernelIdLi5ELi2ELb1ELb1ELb0EEEvNS_12PushPassArgsIT_XT0_EEEPd"
    found = {}
    with tempfile.TemporaryDirectory() as td:
        for f in _gfx950_code_objects(lo, td):
            syms = subprocess.run([readelf, "--syms", "-W", f], capture_output=True, text=True, check=True).stdout
            for name in (fast, slow):
                if name in found or not re.search(r"FUNC\s+\S+\s+\S+\s+\d+\s+" + re.escape(name) + r"$", syms, re.M):
                    continue
                dis = subprocess.run([objdump, "-d", f"--disassemble-symbols={name}", f], capture_output=True, text=True,
                                     check=True).stdout
                run = best = loads = 0
                for line in dis.splitlines():
                    if "global_load" in line:
                        loads += 1
                        run += 1
                        best = max(best, run)
                    elif re.search(r"s_waitcnt.*vmcnt\(0\)", line):
                        run = 0
                notes = subprocess.run([readelf, "--notes", f], capture_output=True, text=True, check=True).stdout
                i = notes.index(name)
                vg = int(re.search(r"\.vgpr_count:\s+(\d+)", notes[i:]).group(1))
                found[name] = (loads, best, vg)
    assert fast in found and slow in found, sorted(found)
    loads, best, vg = found[fast]
    assert best >= (5 + 2) * 2, f"FAST push pass: longest run of loads without a full drain is {best} of {loads}"
    loads, best, vg = found[slow]
    assert vg <= 112, f"HBM-bound push pass (5 columns): {vg} VGPRs — fewer than 4 waves per SIMD"


def test_no_kernel_of_the_library_uses_scratch(lo):
    """Static check on the built gfx950 code objects (no GPU needed): every kernel of libmxlo.so has
    `.private_segment_fixed_size: 0`. A run-time index into a per-lane register array, or one accumulator too many,
    silently moves data to scratch memory — which cost 3x on the single-launch quasi-Newton apply before it was noticed
    (DESIGN §4). The code objects are read out of the HIP fat binary (clang offload bundle) embedded in the library."""
    import os
    import re
    import struct
    import subprocess
    import tempfile
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not os.path.exists(readelf):
        pytest.skip("llvm-readelf not available")
    so = os.path.join(os.path.dirname(lo._lib.__file__), "csrc", "libmxlo.so")
    blob = open(so, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    nkern, spilled = 0, []
    with tempfile.TemporaryDirectory() as td:
        for bi, m in enumerate(re.finditer(re.escape(magic), blob)):
            p = m.start()
            (n,) = struct.unpack_from("<Q", blob, p + 24)
            off = p + 32
            for e in range(n):
                o, size, tl = struct.unpack_from("<QQQ", blob, off)
                off += 24
                triple = blob[off:off + tl].decode()
                off += tl
                if "gfx950" not in triple or size == 0:
                    continue
                f = os.path.join(td, f"co_{bi}_{e}.co")
                with open(f, "wb") as fh:
                    fh.write(blob[p + o:p + o + size])
                notes = subprocess.run([readelf, "--notes", f], capture_output=True, text=True, check=True).stdout
                name = None
                for line in notes.splitlines():
                    mm = re.match(r"\s+\.name:\s+(\S+)", line)
                    if mm:
                        name = mm.group(1)
                    mm = re.match(r"\s+\.private_segment_fixed_size:\s+(\d+)", line)
                    if mm and name:
                        nkern += 1
                        if int(mm.group(1)) > 0:
                            spilled.append((name, int(mm.group(1))))
    assert nkern > 1000, nkern                     # every translation unit was found
    assert not spilled, spilled[:10]


def test_no_gpu_means_loud_failure(lo):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback|no HIP device"):
        lo.opDiagonal(torch.ones(4, dtype=torch.float64))
    with pytest.raises(RuntimeError):
        lo.get_ctx()


def test_product_never_imports_oracle():
    """The product path must not route through the oracle (or any CPU fallback)."""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "linearoperators.jl_amd")
    for dp, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dp, f)).read()
                assert "import oracle" not in text and "lo_oracle" not in text, f


def test_jrange_matches_julia_ranges(lo):
    assert list(lo.jrange(3, 6).to_numpy()) == [3, 4, 5, 6]
    assert list(lo.jrange(1, 7, 2).to_numpy()) == [1, 3, 5, 7]
    assert list(lo.jrange(1, 8, 2).to_numpy()) == [1, 3, 5, 7]
    assert list(lo.jrange(7, 1, -3).to_numpy()) == [7, 4, 1]
    assert len(lo.jrange(5, 4)) == 0
    with pytest.raises(ValueError):
        lo.jrange(1, 2, 0)


def test_storage_promotion_and_nargs(lo):
    from linearoperators_jl_amd.operators import _nargs, promote_storage
    S64 = lo.Storage(torch.float64, torch.device("cuda", 0))
    S32 = lo.Storage(torch.float32, torch.device("cuda", 0))
    Sc = lo.Storage(torch.float64, torch.device("cpu"))
    assert promote_storage(S64, S32) == S64
    with pytest.raises(lo.LinearOperatorException, match="cannot be promoted"):
        promote_storage(S64, Sc)                     # src/operations.jl:138-147
    assert _nargs(lambda res, v, a, b: None) == 4 and _nargs(lambda res, v: None) == 2
    from linearoperators_jl_amd.operators import scalar_flags
    assert scalar_flags(torch.float32, 2.0, 0.0) == lo._lib.SCALARS_F64
    assert scalar_flags(torch.float32, np.float32(2), 0.5) == lo._lib.BETA_F64      # each scalar on its own
    assert scalar_flags(torch.float32, 2.0, np.float32(0.5)) == lo._lib.ALPHA_F64
    assert scalar_flags(torch.float32, np.float32(2), np.float32(0)) == 0
    assert scalar_flags(torch.float32, 2, 0) == 0    # Julia Int scalars never widen a Float32 product
    assert scalar_flags(torch.float64, 2.0, 0.0) == 0


def test_wrapper_algebra_without_device(lo):
    """adjoint/transpose/conj identities (src/adjtrans.jl:33-45) need no device."""
    S = lo.Storage(torch.float64, torch.device("cuda", 0))
    op = lo.LinearOperator(torch.float64, 3, 5, False, False, lambda r, v, a, b: None, None, None, S=S)
    assert lo.adjoint(lo.adjoint(op)) is op and lo.transpose(lo.transpose(op)) is op and lo.conj(lo.conj(op)) is op
    assert isinstance(lo.adjoint(lo.transpose(op)), lo.ConjugateLinearOperator)
    assert lo.adjoint(lo.conj(op)).parent is op and isinstance(lo.adjoint(lo.conj(op)), lo.TransposeLinearOperator)
    assert op.T.shape == (5, 3) and op.H.size(1) == 5 and lo.conj(op).shape == (3, 5)
    with pytest.raises(lo.LinearOperatorException):
        op.size(3)
    with pytest.raises(lo.LinearOperatorException, match="shape mismatch"):
        lo.mul(torch.empty(3), op, torch.empty(4))   # the check fires before any data is touched
    with pytest.raises(lo.LinearOperatorException, match="unable to infer"):
        lo.mul(torch.empty(5), op.T, torch.empty(3))


def test_shard_plan(lo):
    P = lo.sharded.ShardPlan
    for n, w in ((10, 3), (400_000_000, 8), (7, 8), (0, 2), (5, 1)):
        rs = P(n, w).ranges()
        assert rs[0][0] == 0 and rs[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
        sizes = [hi - lo_ for lo_, hi in rs]
        assert max(sizes) - min(sizes) <= 1
    blocks, rows = P(0, 4).blocks_to_ranks([1024] * 1024)
    assert [b - a for a, b in blocks] == [256] * 4 and rows[1] == (256 * 1024, 512 * 1024)
    blocks, rows = P(0, 8).blocks_to_ranks([5, 1, 1, 1])
    assert sum(b - a for a, b in blocks) == 4


def test_sorted_scatter_plan_is_last_write_wins(lo):
    """The sorted plan behind the segment-owner extension kernel equals the sequential `res .= 0; res[I] = u`
    (src/special-operators.jl:171-174) for any index list: sorted, unsorted, with duplicates, empty."""
    from linearoperators_jl_amd.leaves import sorted_scatter_plan
    rng = np.random.default_rng(0)
    cases = [np.array([], dtype=np.int64), np.array([5]), np.arange(1, 50), np.array([2, 5, 2, 3, 5, 5]),
             rng.permutation(40)[:17] + 1, rng.integers(1, 30, 100), np.sort(rng.integers(1, 30, 100))]
    for idx in cases:
        n = 64
        u = rng.standard_normal(idx.size)
        want = np.zeros(n)
        for k, i in enumerate(idx):                      # the reference's sequential assignment
            want[i - 1] = u[k]
        svals, spos = sorted_scatter_plan(idx)
        assert np.all(np.diff(svals) > 0)                # strictly increasing: what mxlo_scatter_zero_sorted requires
        got = np.zeros(n)
        got[svals - 1] = u[spos] if spos is not None else u
        assert np.array_equal(got, want)
        if idx.size and np.all(np.diff(idx) > 0):
            assert spos is None and svals is not None and np.array_equal(svals, idx)


def test_stored_colmajor_aliases_both_layouts(lo):
    from linearoperators_jl_amd.leaves import _stored_colmajor
    M = torch.arange(12.0).reshape(3, 4)                 # row-major: aliased as the column-major transpose
    St, tr = _stored_colmajor(M)
    assert tr and St.data_ptr() == M.data_ptr() and St.shape == (4, 3) and St.stride(0) == 1
    Mc = M.t().contiguous().t()                          # column-major
    St, tr = _stored_colmajor(Mc)
    assert not tr and St.data_ptr() == Mc.data_ptr()
    V = torch.arange(40.0).reshape(4, 10)[:, ::2]        # neither: copied once to column-major
    St, tr = _stored_colmajor(V)
    assert not tr and St.stride(0) == 1 and torch.equal(St, V)


def test_universal_identity_operator(lo):
    """test/test_linop.jl:260-288 "Identity (non-convertible to matrix)": opEye() * x === x for vectors, matrices and
    operators, on either side; opEye() === opEye(); adjoint / transpose / conj return it unchanged."""
    op = lo.opEye()
    v = torch.tensor([1.0, -1.0, 1.0, -1.0, 1.0])
    assert (op * v) is v and (v * op) is v
    A1 = torch.rand(5, 5)
    assert (op * A1) is A1 and (A1 * op) is A1
    T1 = lo.LinearOperator(torch.float64, 5, 5, False, False, lambda res, x, a, b: None, None, None,
                           S=lo.Storage(torch.float64, torch.device("cpu")))
    assert (op * T1) is T1 and (T1 * op) is T1
    op2 = lo.opEye()
    assert op is op2 and (op * op2) is op and (op2 * op) is op
    assert lo.transpose(op) is op and lo.adjoint(op) is op and lo.conj(op) is op and op.T is op and op.H is op
    assert (lo.transpose(op) * v) is v and (lo.conj(op) * v) is v
    assert repr(op) == "Identity operator"


def test_mul_on_matrices_routing_without_device(lo):
    """`mul!(res::AbstractMatrix, op, m::AbstractMatrix, α, β)` — src/operations.jl:34-36 hands the matrices to the closure
    as they are (no shape check, no counter); src/adjtrans.jl:139-156, 207-224: the wrappers check shapes, go to the parent
    when it is hermitian / symmetric, else to ctprod! / tprod!, else "Not implemented"; :251-261 conj.(m) ... conj!(res).
    Host logic only: closures written in Python on CPU tensors."""
    S = lo.Storage(torch.float64, torch.device("cpu"))
    M = torch.tensor([[1.0, 2.0, 0.5], [-1.0, 0.0, 3.0]])                                  # 2 x 3

    def prod(res, v, a, b):
        res.copy_(a * (M @ v) + (b * res if b != 0 else 0))

    def tprod(res, u, a, b):
        res.copy_(a * (M.t() @ u) + (b * res if b != 0 else 0))

    op = lo.LinearOperator(torch.float64, 2, 3, False, False, prod, tprod, None, S=S)
    mv, mu = torch.tensor([[1.0, -2.0], [-1.0, 2.0], [1.0, -2.0]]), torch.tensor([[1.0, -2.0], [-1.0, 2.0]])
    res = torch.empty(2, 2)
    lo.mul(res, op, mv)
    assert torch.equal(res, M @ mv) and lo.nprod(op) == 0                                  # no counter on this path
    rt = torch.ones(3, 2)
    lo.mul(rt, op.T, mu, 2.0, -1.0)
    assert torch.equal(rt, 2.0 * (M.t() @ mu) - 1.0)
    with pytest.raises(lo.LinearOperatorException, match="Not implemented"):
        lo.mul(torch.empty(3, 2), op.H, mu)                                               # no ctprod!, not hermitian
    with pytest.raises(lo.LinearOperatorException, match="shape mismatch"):
        lo.mul(torch.empty(3, 3), op.T, mu)
    sym = lo.LinearOperator(torch.float64, 2, 2, True, True, lambda r, v, a, b: r.copy_(a * v), None, None, S=S)
    r2 = torch.empty(2, 2)
    lo.mul(r2, sym.H, mu, 3.0, 0.0)                                                        # hermitian parent: mul!(res, p, m, α, β)
    assert torch.equal(r2, 3.0 * mu)
    lo.mul(r2, lo.conj(sym), mu, 1.0, 0.0)                                                 # real data: conj is the identity
    assert torch.equal(r2, mu)



def test_columnwise_closures_on_matrices_without_device(lo):
    """The column loop behind `mul!` on matrices for closures marked `columnwise` (what the device closures of dense /
    diagonal / identity operators are): Julia-layout operands are used in place, a row-major `res` is written back, the
    column counts must agree, and an unmarked closure of the package refuses matrices. CPU tensors, Python closures."""
    from linearoperators_jl_amd import operators as ops
    S = lo.Storage(torch.float64, torch.device("cpu"))
    d = torch.tensor([2.0, -1.0, 0.5])
    seen = []

    def diag(res, v, a, b):
        assert res.dim() == 1 and v.dim() == 1 and res.stride(0) == 1 and v.stride(0) == 1     # contiguous columns
        seen.append(res.data_ptr())
        res.copy_(a * d * v + (b * res if b != 0 else 0))

    op = lo.LinearOperator(torch.float64, 3, 3, True, True, ops.columnwise(diag), None, None, S=S)
    m = torch.tensor([[1.0, 4.0], [2.0, 5.0], [3.0, 6.0]])                                   # row-major (torch default)
    for res in (torch.ones(3, 2), torch.ones(2, 3).t()):                                     # row-major, column-major
        seen.clear()
        lo.mul(res, op, m, 2.0, -1.0)
        assert torch.equal(res, 2.0 * d[:, None] * m - 1.0)
        if res.stride(0) == 1:                                                               # Julia layout: written in place
            assert seen == [res[:, 0].data_ptr(), res[:, 1].data_ptr()]
    lo.mul(res, op.H, m)                                                                     # hermitian parent: same closure
    assert torch.equal(res, d[:, None] * m)
    with pytest.raises(lo.LinearOperatorException, match="shape mismatch"):
        lo.mul(torch.empty(3, 3), op, m)
    plain = lo.LinearOperator(torch.float64, 3, 3, True, True, ops.sum_prod, None, None, S=S)   # a closure of the package, unmarked
    with pytest.raises(lo.LinearOperatorException, match="vectors only"):
        lo.mul(torch.empty(3, 2), plain, m)


def test_matrix_operands_with_wrong_row_counts_are_refused_before_any_launch(lo):
    """`mul!` on matrices with a base operator must compare the ROW counts of `res` and `m` with the
    operator: the device closures take raw pointers and their sizes from the operator, so the reference's
    DimensionMismatch (BLAS / broadcast) has to be raised by the host mirror. CPU tensors, a closure that must not run."""
    from linearoperators_jl_amd import operators as ops
    S = lo.Storage(torch.float64, torch.device("cpu"))
    ran = []
    op = lo.LinearOperator(torch.float64, 3, 4, False, False, ops.columnwise(lambda r, v, a, b: ran.append(1)), None, None, S=S)
    for res, m in ((torch.empty(3, 2), torch.empty(5, 2)),       # m has 5 rows, op has 4 columns
                   (torch.empty(2, 2), torch.empty(4, 2)),       # res has 2 rows, op has 3
                   (torch.empty(3, 2), torch.empty(4, 3))):      # column counts differ
        with pytest.raises(lo.LinearOperatorException, match="shape mismatch"):
            lo.mul(res, op, m, 1.0, 0.0)
    assert not ran
    lo.mul(torch.empty(3, 2), op, torch.empty(4, 2), 1.0, 0.0)
    assert ran == [1, 1]


def test_neg_scale_and_sum_accept_matrices_like_the_reference(lo):
    """`-op`, `x*op`, `op1+op2` call mul!(res::AbstractVecOrMat, …) in the reference (src/operations.jl:103-105,
    165-167, 187-197), so they take matrices; `op1*op2` (vector temporaries) does not — in the reference either."""
    from linearoperators_jl_amd import operators as ops
    S = lo.Storage(torch.float64, torch.device("cpu"))
    d1, d2 = torch.tensor([2.0, -1.0, 0.5]), torch.tensor([1.0, 3.0, -2.0])

    def diag(d):
        return lo.LinearOperator(torch.float64, 3, 3, True, True,
                                 ops.columnwise(lambda r, v, a, b: r.copy_(a * d * v + (b * r if b != 0 else 0))), None, None, S=S)
    A, B = diag(d1), diag(d2)
    m = torch.tensor([[1.0, 4.0], [2.0, 5.0], [3.0, 6.0]]).t().contiguous().t()      # Julia layout
    res = torch.ones(2, 3).t()
    lo.mul(res, -A, m, 2.0, -1.0)
    assert torch.equal(res, -2.0 * d1[:, None] * m - 1.0)
    lo.mul(res, 3.0 * A, m, 1.0, 0.0)
    assert torch.equal(res, 3.0 * d1[:, None] * m)
    lo.mul(res, A + B, m, 2.0, 0.0)
    assert torch.equal(res, 2.0 * (d1 + d2)[:, None] * m)
    lo.mul(res, (A - B).T, m)                                                         # symmetric parent: same closure
    assert torch.equal(res, (d1 - d2)[:, None] * m)
    with pytest.raises(lo.LinearOperatorException, match="vectors only"):
        lo.mul(res, A * B, m)


def test_sparse_chunk_tables_cover_every_entry_once(lo):
    """Host logic of the sparse leaf (no device): the work decomposition `mxlo_csc_create` builds for a compressed-row
    operand. For random row-length distributions — empty rows, short rows, rows around the long-row threshold (512),
    rows around and far beyond a chunk (2048) — every stored entry lies in exactly one chunk, chunks are in entry order,
    whole-row chunks hold <= 2048 entries and <= 2048 rows and no row above 512 entries, a row of 513 … 2048 entries is
    a chunk of its own, a longer row is cut into consecutive 2048-entry pieces with consecutive carry slots."""
    import ctypes as C
    L = C.CDLL(lo._lib.LIB_PATH)
    f = L.mxlo_debug_csc_chunks
    f.argtypes = [C.POINTER(C.c_int64), C.c_int64, C.POINTER(C.c_int64), C.c_int64] + [C.POINTER(C.c_int64)] * 3
    f.restype = C.c_int32
    rng = np.random.default_rng(7)
    CH, LONG = 2048, 512
    for trial in range(40):
        nrows = int(rng.integers(0, 3000))
        kind = trial % 5
        if kind == 0:
            lens = rng.integers(0, 12, nrows)
        elif kind == 1:
            lens = rng.choice([0, 1, 7, 500, 512, 513, 2047, 2048, 2049, 5000, 10000], nrows)
        elif kind == 2:
            lens = np.zeros(nrows, np.int64)
        elif kind == 3:
            lens = rng.integers(400, 700, nrows)
        else:
            lens = (rng.pareto(1.2, nrows) * 20).astype(np.int64)
        ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        nchunks, nlong, ncarry = C.c_int64(), C.c_int64(), C.c_int64()
        assert f(ptr.ctypes.data_as(C.POINTER(C.c_int64)), nrows, None, 0, C.byref(nchunks), C.byref(nlong), C.byref(ncarry)) == 0
        out = np.zeros((max(1, nchunks.value), 6), np.int64)
        assert f(ptr.ctypes.data_as(C.POINTER(C.c_int64)), nrows, out.ctypes.data_as(C.POINTER(C.c_int64)), out.shape[0],
                 C.byref(nchunks), C.byref(nlong), C.byref(ncarry)) == 0
        out = out[:nchunks.value]
        pos, next_row, carry = 0, 0, 0
        for k0, nz, row0, nr, ck, cs in out:
            assert k0 == pos and nz <= CH, (trial, k0, pos)
            if ck == 0:                                               # whole rows
                assert row0 == next_row and 1 <= nr <= CH
                assert ptr[row0] == k0 and ptr[row0 + nr] == k0 + nz
                assert (np.diff(ptr[row0:row0 + nr + 1]) <= LONG).all()
                next_row = row0 + nr
            elif ck == 1:                                             # one long row
                assert row0 == next_row and nr == 1 and LONG < nz <= CH and ptr[row0] == k0 and ptr[row0 + 1] == k0 + nz
                next_row = row0 + 1
            else:                                                     # a piece of a row beyond a chunk
                assert ck == 2 and nr == 1 and cs == carry and ptr[row0 + 1] - ptr[row0] > CH
                assert ptr[row0] <= k0 and k0 + nz <= ptr[row0 + 1]
                assert nz == CH or k0 + nz == ptr[row0 + 1]           # only the last piece is short
                carry += 1
                if k0 + nz == ptr[row0 + 1]:
                    assert row0 == next_row
                    next_row = row0 + 1
            pos += nz
        assert pos == ptr[-1] and next_row == nrows and carry == ncarry.value
        assert nlong.value == int((np.diff(ptr) > CH).sum())
