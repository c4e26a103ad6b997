"""Row-sharded vectors across the GPUs of one node: one process per GPU, `torch.distributed`
(backend "nccl" == RCCL over xGMI) for the only exchange the hot path has — the sum of partial dots.

The reference has no distributed path (no NCCL/MPI call site); this module is the seam
north_star asks for. Every operator of this package works unchanged on its local row range
`[lo, hi)`; the *only* cross-rank data are the scalars produced by global reductions (Householder
h'v, L-BFGS / L-SR1 panel dots, push! dots, shifted-solve dots). libmxlo.so exposes one hook for
all of them (`mxlo_ctx_set_allreduce`, include/mxlo.h): after the local fixed-order finalize it
hands over `count` doubles in device memory to be sum-all-reduced in place, stream-ordered.
All ranks then hold bit-identical scalars (RCCL all-reduce returns the same bits everywhere),
which is what keeps the replicated control flow (`ys[k] != 0` skips, push! rejection) consistent.

Message sizes are tiny (8 B for Householder, 16*m B for a forward L-BFGS apply): latency-bound, so
nothing here is bucketed. Paths that do not shard (kron at 1024², general restriction) run as
replicas: see DESIGN.md.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch
import torch.distributed as dist


@dataclass(frozen=True)
class ShardPlan:
    """Contiguous row ranges: rank r owns rows [lo(r), hi(r)) of an n-vector; the first n % world
    ranks own one extra row (so shards differ by at most one row)."""
    n: int
    world: int

    def lo(self, rank: int) -> int:
        q, r = divmod(self.n, self.world)
        return rank * q + min(rank, r)

    def hi(self, rank: int) -> int:
        return self.lo(rank + 1) if rank + 1 < self.world else self.n

    def local_n(self, rank: int) -> int:
        return self.hi(rank) - self.lo(rank)

    def ranges(self):
        return [(self.lo(r), self.hi(r)) for r in range(self.world)]

    def blocks_to_ranks(self, block_rows):
        """BlockDiagonalOperator sharding: whole blocks per rank by cumulative row
        offset — a block belongs to the rank that owns its first row. Returns a list of
        (first_block, last_block_exclusive) per rank and the induced row ranges."""
        offs = np.concatenate([[0], np.cumsum(np.asarray(block_rows, dtype=np.int64))])
        total = int(offs[-1])
        owner = np.minimum((offs[:-1] * self.world) // max(total, 1), self.world - 1)
        out, rows = [], []
        for r in range(self.world):
            ids = np.nonzero(owner == r)[0]
            if ids.size:
                out.append((int(ids[0]), int(ids[-1]) + 1))
                rows.append((int(offs[ids[0]]), int(offs[ids[-1] + 1])))
            else:
                out.append((0, 0))
                rows.append((0, 0))
        return out, rows


class _DevArray:
    """Minimal __cuda_array_interface__ carrier so torch can alias a raw device pointer."""

    def __init__(self, p: int, count: int):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (p, False), "version": 2,
                                         "strides": None}


def wrap_doubles(p: int, count: int, cuda: bool) -> torch.Tensor:
    """Alias `count` doubles at address `p` (device or host memory) as a 1-D float64 tensor."""
    if cuda:
        return torch.as_tensor(_DevArray(p, count), device="cuda")
    arr = np.ctypeslib.as_array((C.c_double * count).from_address(p))
    return torch.from_numpy(arr)


def make_allreduce_hook(group=None, cuda: bool = True):
    """The Python body of `mxlo_allreduce_fn`: sum-all-reduce `count` doubles in place."""
    cache: dict = {}

    def hook(user, dev_buf, count, stream):
        try:
            key = (int(dev_buf), int(count))
            t = cache.get(key)
            if t is None:
                t = cache[key] = wrap_doubles(int(dev_buf), int(count), cuda)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            return 0
        except Exception as e:  # never let an exception cross the C ABI
            import sys
            print(f"[mxlo all-reduce hook] {e!r}", file=sys.stderr)
            return 1

    return hook


class NativeRcclHook:
    """RCCL communicator owned by libmxlo_rccl.so; the hook is the C function `mxlo_rccl_allreduce_hook`
    itself (ncclAllReduce on the ctx stream) — no Python, no extra stream, nothing synchronises."""

    def __init__(self, rank: int, world: int, group=None, timeout_s: float = 120.0, unique_id: bytes | None = None):
        """`unique_id`: the 128-byte id when the host runtime has already delivered it (MPI, a file, a test); otherwise
        rank 0 creates it and it is broadcast over torch.distributed."""
        from . import _lib
        R = _lib.rccl_lib()
        idbuf = torch.zeros(_lib.RCCL_ID_BYTES, dtype=torch.uint8)
        if unique_id is not None:
            idbuf = torch.tensor(list(unique_id[:_lib.RCCL_ID_BYTES].ljust(_lib.RCCL_ID_BYTES, b"\0")), dtype=torch.uint8)
        elif rank == 0:
            raw = (C.c_ubyte * _lib.RCCL_ID_BYTES)()
            if R.mxlo_rccl_unique_id(raw) != 0:
                raise RuntimeError(R.mxlo_rccl_last_error().decode())
            idbuf = torch.tensor(list(raw), dtype=torch.uint8)
        if world > 1 and unique_id is None:             # the 128-byte id travels over the host runtime
            dev = torch.device("cuda", torch.cuda.current_device())
            backend = dist.get_backend(group)
            t = idbuf.to(dev) if backend == "nccl" else idbuf
            dist.broadcast(t, src=0, group=group)
            idbuf = t.cpu()
        raw = (C.c_ubyte * _lib.RCCL_ID_BYTES)(*idbuf.tolist())
        self.comm = C.c_void_p()
        # ncclCommInitRank is collective and cannot be interrupted: run it on a helper thread (HIP's current
        # device is per thread) and give up after `timeout_s` so that a broken bootstrap degrades to the
        # torch.distributed transport instead of hanging the job.
        import threading
        dev_index = torch.cuda.current_device()
        out = {}

        def build():
            torch.cuda.set_device(dev_index)
            out["rc"] = R.mxlo_rccl_comm_create(rank, world, raw, C.byref(self.comm))
            out["err"] = R.mxlo_rccl_last_error().decode()

        th = threading.Thread(target=build, daemon=True)
        th.start()
        th.join(timeout_s)
        if th.is_alive():
            raise TimeoutError(f"ncclCommInitRank did not return within {timeout_s:.0f} s")
        if out.get("rc", 1) != 0:
            raise RuntimeError(out.get("err", "mxlo_rccl_comm_create failed"))
        self._R = R
        self.fn = C.cast(R.mxlo_rccl_allreduce_hook, _lib.ALLREDUCE_FN)

    def install(self, ctx):
        from . import _lib
        ctx._hook = self                                 # keep the communicator alive with the ctx
        _lib.call("mxlo_ctx_set_allreduce", ctx.handle, self.fn, self.comm)

    def info(self) -> dict:
        """What the communicator itself reports (ncclCommCount / ncclCommUserRank / ncclCommCuDevice + PCI bus id)."""
        ranks, ur, dv = C.c_int32(-1), C.c_int32(-1), C.c_int32(-1)
        pci = C.create_string_buffer(64)
        if self._R.mxlo_rccl_comm_info(self.comm, C.byref(ranks), C.byref(ur), C.byref(dv), pci, 64) != 0:
            raise RuntimeError(self._R.mxlo_rccl_last_error().decode())
        return {"ranks_seen": ranks.value, "user_rank": ur.value, "device": dv.value, "pci_bus_id": pci.value.decode()}

    def preflight(self, stream: int, reps: int = 50, timeout_ms: int = 60000) -> dict:
        """Collective: known-answer + identical-bits all-reduces of 8 B / 320 B / 6912 B through the native hook, agreed
        verdict, latency in us (include/mxlo_rccl.h: mxlo_rccl_preflight). Raises RuntimeError naming phase and rank."""
        lat = (C.c_double * 3)()
        if self._R.mxlo_rccl_preflight(self.comm, C.c_void_p(stream), reps, timeout_ms, lat) != 0:
            raise RuntimeError(self._R.mxlo_rccl_last_error().decode())
        return {"8B": round(lat[0], 2), "320B": round(lat[1], 2), "6912B": round(lat[2], 2)}

    def abort(self):
        if self.comm:
            self._R.mxlo_rccl_comm_abort(self.comm)
            self.comm = C.c_void_p()

    def close(self):
        if self.comm:
            self._R.mxlo_rccl_comm_destroy(self.comm)
            self.comm = C.c_void_p()


class PeerShmHook:
    """The peer-mapped one-shot exchange for one process per GPU (include/mxlo_rccl.h: mxlo_peer_comm_create_shm):
    mailboxes in a POSIX shared-memory segment registered with every rank's HIP runtime; the hook is the C function
    `mxlo_peer_allreduce_hook` (ONE kernel per collective on the ctx stream, fixed rank order: identical bits on all
    ranks). The segment name travels over the host runtime (torch.distributed object broadcast)."""

    def __init__(self, rank: int, world: int, group=None, timeout_ms: int = 30000):
        from . import _lib
        import os
        R = _lib.rccl_lib()
        name = [f"/mxlo-{os.getpid()}-{int.from_bytes(os.urandom(6), 'little'):x}" if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(name, src=0, group=group)
        self.name, self.rank, self.world = name[0], rank, world
        self.comm = C.c_void_p()
        err = None
        if rank == 0 and R.mxlo_peer_comm_create_shm(self.name.encode(), 0, world, 1, timeout_ms, C.byref(self.comm)) != 0:
            err = R.mxlo_peer_last_error().decode()
        if world > 1:
            dist.barrier(group=group)                     # the segment exists (or rank 0 failed: the others fail to open it)
        if rank != 0 and R.mxlo_peer_comm_create_shm(self.name.encode(), rank, world, 0, timeout_ms, C.byref(self.comm)) != 0:
            err = R.mxlo_peer_last_error().decode()
        if err is not None:
            raise RuntimeError(err)
        self._R = R
        self.fn = C.cast(R.mxlo_peer_allreduce_hook, _lib.ALLREDUCE_FN)
        # the communicator owns a /dev/shm segment (rank 0 its NAME): released with the object even when nobody calls
        # close() — a failed agreement round, an exception between construction and tear-down
        import weakref
        self._fin = weakref.finalize(self, PeerShmHook._release, R, self.comm)

    @staticmethod
    def _release(R, comm):
        if comm:
            R.mxlo_peer_comm_destroy(comm)
            comm.value = None

    def install(self, ctx):
        from . import _lib
        ctx._hook = self
        _lib.call("mxlo_ctx_set_allreduce", ctx.handle, self.fn, self.comm)

    def preflight(self, stream: int, reps: int = 50, timeout_ms: int = 60000) -> dict:
        lat = (C.c_double * 3)()
        fn = C.cast(self._R.mxlo_peer_allreduce_hook, C.c_void_p)
        if self._R.mxlo_rccl_preflight_hook(fn, self.comm, None, self.rank, self.world, C.c_void_p(stream), reps, timeout_ms, lat) != 0:
            raise RuntimeError(self._R.mxlo_rccl_last_error().decode())
        return {"8B": round(lat[0], 2), "320B": round(lat[1], 2), "6912B": round(lat[2], 2)}

    def check(self):
        """After a stream synchronisation: raises if a gather of this rank timed out (its results are NaN)."""
        if self._R.mxlo_peer_comm_check(self.comm) != 0:
            raise RuntimeError(self._R.mxlo_peer_last_error().decode())

    def debug(self, key: str, value: int):
        if self._R.mxlo_peer_comm_debug(self.comm, key.encode(), int(value)) != 0:
            raise RuntimeError(self._R.mxlo_peer_last_error().decode())

    def close(self):
        self._fin()                                       # destroys the communicator once (detaches the finalizer)


def install_allreduce(ctx, group=None, native: bool = True):
    """Route every global reduction of `ctx` through an RCCL all-reduce (no-op when world == 1).

    native=True  : libmxlo_rccl.so — ncclAllReduce issued from C on the ctx stream (default);
    native=False : the Python hook over torch.distributed (also what the gloo CPU tests exercise)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        ctx.set_allreduce(None)
        return None
    if native:
        hook = NativeRcclHook(dist.get_rank(group), dist.get_world_size(group), group)
        hook.install(ctx)
        return hook
    ctx.set_allreduce(make_allreduce_hook(group, cuda=True))
    return None


def install_agreed_allreduce(ctx, group=None, timeout_s: float = 120.0, require_native: bool = False,
                             even_at_world_1: bool = False):
    """The native RCCL transport when EVERY rank managed to build its communicator, otherwise the
    torch.distributed hook on every rank (never a mix: mixed transports would deadlock). Returns the
    NativeRcclHook or None. `require_native=True` (what `bench.py --gpus N` uses): no degraded transport —
    if any rank failed, EVERY rank raises RuntimeError after the agreement round. `even_at_world_1`: build and install the
    communicator for a group of ONE rank too (a rehearsal of the N-rank path on a one-GPU box)."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not even_at_world_1):
        ctx.set_allreduce(None)
        return None
    hook, ok = None, 1
    try:
        hook = NativeRcclHook(dist.get_rank(group), dist.get_world_size(group), group, timeout_s)
    except Exception as e:
        import sys
        print(f"[mxlo] native RCCL transport unavailable on rank {dist.get_rank(group)}: {e!r}", file=sys.stderr)
        ok = 0
    on_gpu = dist.get_backend(group) == "nccl"
    flag = torch.tensor([ok], dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device()) if on_gpu else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    if int(flag.item()) == 1:
        hook.install(ctx)
        return hook
    if require_native:
        raise RuntimeError("the native RCCL all-reduce hook (libmxlo_rccl.so) could not be built on every rank "
                           f"(rank {dist.get_rank(group)}: {'ok' if ok else 'FAILED'}); refusing the Python-issued transport")
    ctx.set_allreduce(make_allreduce_hook(group, cuda=True))
    return None


def uninstall_allreduce(ctx) -> None:
    ctx.set_allreduce(None)


def shard(full: torch.Tensor, plan: ShardPlan, rank: int, device=None) -> torch.Tensor:
    """Local row range of a host/global vector (test & data-loading helper)."""
    t = full[plan.lo(rank):plan.hi(rank)].contiguous()
    return t.to(device) if device is not None else t


# ---------------------------------------------------------------------------------------------------------------
# the one hot-path piece that moves VECTORS (not scalars) between GPUs — a dense `LinearOperator(M)`
# whose rows are sharded like every vector. Rank r holds M[lo_m(r):hi_m(r), :] (column-major, all n columns),
# the input shard v[lo_n(r):hi_n(r)] and the output shard res[lo_m(r):hi_m(r)]:
#   prod!  : all-gather(v) over xGMI, then the local GEMV (mxlo_gemv N) with the caller's α, β;
#   tprod! : local Mᵀ·u (a full n-vector of partial sums), reduce-scatter(sum), then res = α·(·) + β·res.
# Ring collectives move (world-1)/world · 8n B per GPU per apply against 8·m_loc·n B of HBM traffic, so the apply
# stays HBM-bound as long as m_loc ≫ 7·(HBM rate / xGMI link rate) ≈ 7·8000/153 ≈ 370 rows per GPU.
# Shards differ by at most one row, so collectives run on buffers padded to the largest shard.
class VectorExchange:
    """The two collectives of the row-sharded dense operators, on the ShardPlan wire format (`world` slots of
    pad = ceil(n / world) elements). Staging is ONE libmxlo launch per direction (`mxlo_shard_stage`; own-slot
    placement is the fused zero-fill + copy `mxlo_scatter_zero_range`) — no per-shard torch copies.

    backend "nccl" (RCCL over xGMI, the product transport): all_gather_into_tensor / reduce_scatter_tensor.
    backend "gloo" (debug transport, two ranks on one GPU): gloo only implements broadcast and all_reduce on device
    tensors, so both collectives are expressed through all_reduce — gather = sum of slot vectors that are zero
    outside the owner's slot (exact), reduce-scatter = all_reduce of the whole padded vector, keep my slot."""

    def __init__(self, plan: ShardPlan, dtype, device, group=None):
        self.plan, self.group = plan, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if plan.world != self.world:
            raise ValueError("shard plan was built for a different world size")
        self.native = dist.get_backend(group) == "nccl"
        self.dt, self.dev = dtype, device
        self.es = torch.empty(0, dtype=dtype).element_size()
        self.n, self.pad = plan.n, -(-plan.n // self.world)
        self.n_loc = plan.local_n(self.rank)
        self.wire = torch.empty(self.world * self.pad, dtype=dtype, device=device)       # all-gather target / reduce input
        self.slot = torch.empty(self.pad, dtype=dtype, device=device)                     # my slot (send / receive)
        self.full = torch.empty(self.n, dtype=dtype, device=device)

    def _stage(self, dst, src, nvalid, direction):
        from . import _lib
        from .device import get_ctx, ptr
        ctx = get_ctx(self.dev)
        _lib.call("mxlo_shard_stage", ctx.handle, self.es, ptr(dst), ptr(src), self.n, self.world, int(nvalid), direction)

    def _place(self, dst, first, v):                     # dst .= 0; dst[first : first + len(v)] = v   (one launch)
        from . import _lib
        from .device import get_ctx, ptr
        ctx = get_ctx(self.dev)
        _lib.call("mxlo_scatter_zero_range", ctx.handle, self.es, ptr(dst), dst.numel(), ptr(v), first + 1, 1, v.numel())

    def all_gather(self, v_local: torch.Tensor) -> torch.Tensor:
        """The whole n-vector (rows in global order) from every rank's shard; returns an internal buffer."""
        from . import _lib
        if self.native:
            self._place(self.slot, 0, v_local)
            dist.all_gather_into_tensor(self.wire, self.slot, group=self.group)
        else:
            self._place(self.wire, self.rank * self.pad, v_local)
            dist.all_reduce(self.wire, op=dist.ReduceOp.SUM, group=self.group)
        self._stage(self.full, self.wire, -1, _lib.SHARD_UNPACK)
        return self.full

    def reduce_scatter(self, partial_full: torch.Tensor, nvalid: int = -1) -> torch.Tensor:
        """My shard of the sum over ranks of `partial_full` (n-vector; rows >= nvalid count as zeros and are not read).
        Returns a view of length n_loc into an internal buffer."""
        from . import _lib
        self._stage(self.wire, partial_full, nvalid, _lib.SHARD_PACK)
        if self.native:
            dist.reduce_scatter_tensor(self.slot, self.wire, op=dist.ReduceOp.SUM, group=self.group)
            return self.slot[:self.n_loc]
        dist.all_reduce(self.wire, op=dist.ReduceOp.SUM, group=self.group)
        return self.wire[self.rank * self.pad:self.rank * self.pad + self.n_loc]


def row_sharded_dense(M_local: torch.Tensor, plan_m: ShardPlan, plan_n: ShardPlan, group=None):
    """LinearOperator over the local row block `M_local` (m_loc × n; dense, or sparse: torch.sparse_csc / _csr) of an
    m × n matrix; operates on shards."""
    from . import _lib
    from .device import Storage, dtype_code, get_ctx, ptr
    from .leaves import LinearOperatorFromMatrix
    from .operators import LinearOperator, scalar_flags

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if plan_m.world != world or plan_n.world != world:
        raise ValueError("shard plans were built for a different world size")
    m_loc, n = M_local.shape
    if m_loc != plan_m.local_n(rank) or n != plan_n.n:
        raise ValueError("M_local must hold this rank's rows and all n columns")
    local = LinearOperatorFromMatrix(M_local)               # mxlo_gemv N/T on the local block
    dev, dt = M_local.device, M_local.dtype
    n_loc = plan_n.local_n(rank)
    ex = VectorExchange(plan_n, dt, dev, group)
    partial = torch.empty(n, dtype=dt, device=dev)

    def prod(res, v, a, b):                                  # res_loc = α·M_loc·v + β·res_loc
        local.prod(res, ex.all_gather(v), a, b)

    def tprod(res, u, a, b):                                 # res_loc = α·(Σ_ranks M_locᵀ·u_loc)[shard] + β·res_loc
        local.tprod(partial, u, 1.0, 0.0)
        mine = ex.reduce_scatter(partial)
        ctx = get_ctx(res.device)
        _lib.call("mxlo_eye_mul", ctx.handle, dtype_code(dt), ptr(res), ptr(mine), n_loc, n_loc, float(a),
                  float(b), scalar_flags(dt, a, b))

    return LinearOperator(dt, m_loc, n_loc, False, False, prod, tprod, tprod, S=Storage(dt, dev))


row_sharded_matrix = row_sharded_dense      # dense OR sparse local block: LinearOperatorFromMatrix routes torch.sparse_csc / _csr
                                            # tensors to the sparse leaf (mxlo_csc_*); the collectives are the same


# opHermitian(d, A) with the ROWS of the lower triangle sharded ("needs all-gather(v) + reduce-scatter
# (Lᴴ part)"). Rank r holds rows [lo, hi) of A (column-major, all n columns; only the strict lower triangle is read),
# d[lo:hi], v[lo:hi] and res[lo:hi]:
#   local rows of L split into the rectangle R = L[lo:hi, 0:lo] and the diagonal triangle T = L[lo:hi, lo:hi];
#   res_loc  = d∘v + R·v[0:lo] + (T + Tᵀ)·v[lo:hi]      -> mxlo_gemv N on R, mxlo_hermitian_mul on the diagonal block
#   + the contributions Rᵀ·v[lo:hi] of the ranks BELOW to columns 0:lo  -> mxlo_gemv T on R, reduce-scatter(sum).
# One all-gather of v and one reduce-scatter of an n-vector per apply; α, β applied once at the end.
def row_sharded_hermitian(d_local: torch.Tensor, A_local: torch.Tensor, plan: ShardPlan, group=None):
    from . import _lib
    from .device import Storage, dtype_code, get_ctx, ptr
    from .leaves import LinearOperatorFromMatrix, opHermitian
    from .operators import LinearOperator, mul, scalar_flags

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    n, lo_, hi_ = plan.n, plan.lo(rank), plan.hi(rank)
    m_loc = hi_ - lo_
    if A_local.shape != (m_loc, n) or d_local.numel() != m_loc:
        raise ValueError("A_local must hold this rank's rows (m_loc x n), d_local its part of the diagonal")
    dev, dt = A_local.device, A_local.dtype
    rect = LinearOperatorFromMatrix(A_local[:, :lo_]) if lo_ > 0 else None        # R = L[lo:hi, 0:lo] (fully below the diagonal)
    tri = opHermitian(d_local, A_local[:, lo_:hi_])                               # d∘v + (T + Tᵀ)·v on the diagonal block
    ex = VectorExchange(plan, dt, dev, group)
    partial = torch.empty(n, dtype=dt, device=dev)          # only rows [0, lo) are ever written or read
    acc = torch.empty(m_loc, dtype=dt, device=dev)

    def prod(res, v, a, b):
        vfull = ex.all_gather(v)
        mul(acc, tri, v, 1.0, 0.0)                                  # d∘v + (T + Tᵀ) v_loc
        if rect is not None:
            mul(acc, rect, vfull[:lo_], 1.0, 1.0)                   # + R v[0:lo]
            mul(partial[:lo_], rect.T, v, 1.0, 0.0)                 # Rᵀ v_loc -> columns 0:lo (owned by the ranks above)
        below = ex.reduce_scatter(partial, nvalid=lo_)              # rows >= lo are zeros by definition (not read)
        ctx = get_ctx(res.device)                                   # acc += what the ranks below contribute to my rows
        _lib.call("mxlo_eye_mul", ctx.handle, dtype_code(dt), ptr(acc), ptr(below), m_loc, m_loc, 1.0, 1.0, 0)
        _lib.call("mxlo_eye_mul", ctx.handle, dtype_code(dt), ptr(res), ptr(acc), m_loc, m_loc, float(a), float(b),
                  scalar_flags(dt, a, b))                           # res = α·acc + β·res

    return LinearOperator(dt, m_loc, m_loc, True, True, prod, prod, prod, S=Storage(dt, dev))
