"""-m gpu: the single-process multi-device C ABI of include/mxlo_rccl.h (mxlo_shard_ctx_create + `_sharded` entry
points taking per-device pointer arrays), driven exactly as a Julia host would through ccall — no torch.distributed,
no MPI. Transports covered here:
  * RCCL communicator(s) from ncclCommInitAll over ALL visible devices (1 on the build pool's boxes; the same test
    spans 2/4/8 devices wherever they are visible, e.g. the driver's 8-GPU node);
  * the loopback transport (2 and 5 shards on one GPU): the multi-shard logic and the bit-identical replicated
    scalars, checked on every box.
Results are compared with the UNSHARDED oracle on the concatenated vectors."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def rel(a, b):
    nb = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (nb if nb else 1.0)


def pairs(rng, n, k):
    out = []
    for _ in range(k):
        s = rng.uniform(-1, 1, n)
        out.append((s, s * rng.uniform(0.5, 2.0, n) + 1e-2 * rng.standard_normal(n)))
    return out


class Shards:
    """Per-device row shards of host vectors + the pointer arrays the C ABI takes."""

    def __init__(self, R, sctx, sizes):
        self.R, self.sctx, self.sizes = R, sctx, sizes
        self.devs = [torch.device("cuda", R.mxlo_shard_ctx_device(sctx, i)) for i in range(len(sizes))]
        self.off = np.concatenate([[0], np.cumsum(sizes)])
        self.nloc = (C.c_int64 * len(sizes))(*sizes)

    def put(self, a):
        return [torch.from_numpy(np.ascontiguousarray(a[self.off[i]:self.off[i + 1]])).to(self.devs[i]) for i in range(len(self.sizes))]

    def ptrs(self, ts):
        return (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])

    def get(self, ts):
        self.R.mxlo_shard_ctx_sync(self.sctx)
        return np.concatenate([t.cpu().numpy() for t in ts])


def configs():
    nvis = torch.cuda.device_count()
    out = [("rccl-all-visible", list(range(nvis)))]
    if nvis > 1:
        out.append(("rccl-2", [0, 1]))
    out += [("loopback-2", [0, 0]), ("loopback-5", [0] * 5)]
    return out


@pytest.mark.parametrize("name,ids", configs())
def test_sharded_abi_matches_unsharded_oracle(lo, dev, name, ids):
    R = lo._lib.rccl_lib()
    nd = len(ids)
    sctx = C.c_void_p()
    rc = R.mxlo_shard_ctx_create(nd, (C.c_int32 * nd)(*ids), C.byref(sctx))
    assert rc == 0, R.mxlo_shard_last_error()
    try:
        assert R.mxlo_shard_ctx_ndev(sctx) == nd
        assert bool(R.mxlo_shard_ctx_is_loopback(sctx)) == name.startswith("loopback")
        rng = np.random.default_rng(len(name) + nd)
        n = 40_003
        cuts = np.sort(rng.choice(np.arange(1, n), nd - 1, replace=False)) if nd > 1 else np.array([], dtype=int)
        sizes = np.diff(np.concatenate([[0], cuts, [n]])).astype(int).tolist()      # ragged shards
        sh = Shards(R, sctx, sizes)
        F64 = lo._lib.F64

        # ---- opHouseholder: one scalar all-reduce
        h = rng.standard_normal(n)
        h /= np.linalg.norm(h)
        v, r0 = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
        hs, vs, rs = sh.put(h), sh.put(v), sh.put(r0)
        assert R.mxlo_householder_mul_sharded(sctx, F64, sh.ptrs(rs), sh.ptrs(hs), sh.ptrs(vs), sh.nloc, 2.0, -3.0, 0) == 0, R.mxlo_shard_last_error()
        assert rel(sh.get(rs), oracle.householder_mul(r0.copy(), h, v, 2.0, -3.0)) <= 1e-12
        # ---- opDiagonal: independent shards, bit-exact
        rs = sh.put(r0)
        assert R.mxlo_diag_mul_sharded(sctx, F64, sh.ptrs(rs), sh.ptrs(hs), sh.ptrs(vs), sh.nloc, 1.5, 0.5, 0) == 0
        assert np.array_equal(sh.get(rs), oracle.diag_mul(r0.copy(), h, v, 1.5, 0.5))

        # ---- quasi-Newton operators: push! (incl. a rejected pair), mul!, solve_shifted_system!, diag!, reset!
        for kind, mem, okind in ((lo._lib.QN_LBFGS_INV, 6, "inv"), (lo._lib.QN_LBFGS_FWD, 5, "fwd"), (lo._lib.QN_LSR1, 4, "lsr1"),
                                 (lo._lib.QN_LBFGS_FWD, 40, "fwd")):
            q = C.c_void_p()
            assert R.mxlo_qn_create_sharded(sctx, kind, F64, sh.nloc, mem, 1, 0, 0.99, 10.0, C.byref(q)) == 0, R.mxlo_shard_last_error()
            Bo = oracle.LSR1(n, mem=mem, scaling=True) if okind == "lsr1" else oracle.LBFGS(n, mem=mem, scaling=True, inverse=(okind == "inv"))
            acc = C.c_int32(-1)
            prs = pairs(rng, n, mem + 3)
            prs.insert(2, (prs[0][0], -prs[0][0]))                      # y's < 0: rejected on every shard
            for s_, y_ in prs:
                ss, ys = sh.put(s_), sh.put(y_)
                assert R.mxlo_qn_push_sharded(q, sh.ptrs(ss), sh.ptrs(ys), C.byref(acc)) == 0, R.mxlo_shard_last_error()
                want = Bo.push(s_, y_)
                if okind == "lsr1":
                    assert bool(acc.value) == want
            # the replicated scalars are BIT-IDENTICAL on every shard (they drive replicated control flow)
            sc0, ys0, ax0 = (C.c_double * 5)(), (C.c_double * mem)(), (C.c_double * mem)()
            assert R.mxlo_qn_get_scalars_sharded(q, 0, sc0, ys0, ax0) == 0
            assert int(sc0[0]) == Bo.insert
            for i in range(1, nd):
                sci, ysi, axi = (C.c_double * 5)(), (C.c_double * mem)(), (C.c_double * mem)()
                assert R.mxlo_qn_get_scalars_sharded(q, i, sci, ysi, axi) == 0
                assert bytes(sci)[:32] == bytes(sc0)[:32] and bytes(ysi) == bytes(ys0) and bytes(axi) == bytes(ax0), (name, i)   # [4] = n_local
            x = rng.uniform(-1, 1, n)
            xs, rs = sh.put(x), sh.put(r0)
            assert R.mxlo_qn_mul_sharded(q, sh.ptrs(rs), sh.ptrs(xs), 2.0, -3.0, 0) == 0, R.mxlo_shard_last_error()
            assert rel(sh.get(rs), Bo.mul(r0.copy(), x, 2.0, -3.0)) <= 1e-9, (name, okind, mem)
            if okind == "fwd":
                sigma = 0.25
                bx = Bo.mul(np.empty(n), x, 1.0, 0.0) + sigma * x
                bs, sol = sh.put(bx), sh.put(np.zeros(n))
                assert R.mxlo_qn_solve_shifted_sharded(q, sh.ptrs(sol), sh.ptrs(bs), sigma) == 0, R.mxlo_shard_last_error()
                assert np.allclose(sh.get(sol), x, atol=1e-6, rtol=1e-6)
            if okind != "inv":
                ds = sh.put(np.zeros(n))
                assert R.mxlo_qn_diag_sharded(q, sh.ptrs(ds)) == 0
                assert rel(sh.get(ds), Bo.diag()) <= 1e-9
            assert R.mxlo_qn_reset_sharded(q) == 0
            rs = sh.put(r0)
            assert R.mxlo_qn_mul_sharded(q, sh.ptrs(rs), sh.ptrs(xs), 1.0, 0.0, 0) == 0
            assert np.array_equal(sh.get(rs), x)                        # empty memory: the identity
            assert R.mxlo_qn_destroy_sharded(q) == 0
    finally:
        assert R.mxlo_shard_ctx_destroy(sctx) == 0


def test_shard_ctx_argument_errors(lo, dev):
    R = lo._lib.rccl_lib()
    sctx = C.c_void_p()
    assert R.mxlo_shard_ctx_create(0, None, C.byref(sctx)) == lo._lib.EINVAL
    assert R.mxlo_shard_ctx_create(1, (C.c_int32 * 1)(99), C.byref(sctx)) == lo._lib.EINVAL
    assert b"device 99" in R.mxlo_shard_last_error()
    assert R.mxlo_shard_ctx_create(1, None, C.byref(sctx)) == 0          # NULL ids: devices 0 .. ndev-1
    assert R.mxlo_shard_ctx_device(sctx, 0) == 0
    assert R.mxlo_householder_mul_sharded(sctx, 0, None, None, None, None, 1.0, 0.0, 0) == lo._lib.EINVAL
    assert R.mxlo_shard_ctx_destroy(sctx) == 0
