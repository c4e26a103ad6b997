/* mxlo_rccl.h — native RCCL transport for the row-sharding seam of mxlo.h (libmxlo_rccl.so).
 *
 * libmxlo.so itself has no communication dependency: every global reduction calls the
 * `mxlo_allreduce_fn` hook installed with mxlo_ctx_set_allreduce(). This small companion library
 * provides that hook on top of RCCL (ncclAllReduce over xGMI), so that a sharded apply never leaves
 * native code: one process per GPU, each creates a communicator from a shared 128-byte unique id
 * (exchanged by the host runtime, e.g. torch.distributed or MPI), then installs
 *     mxlo_ctx_set_allreduce(ctx, mxlo_rccl_allreduce_hook, comm);
 * The collective is enqueued on the ctx stream: stream-ordered between the local finalize and the
 * kernel that consumes the scalars, no host synchronisation. The reference has no counterpart
 * (no NCCL/MPI call site); message sizes are 8 B .. a few KB (latency-bound).
 */
#ifndef MXLO_RCCL_H
#define MXLO_RCCL_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MXLO_RCCL_ID_BYTES 128

/* rank 0: fill `id_out` (MXLO_RCCL_ID_BYTES bytes) with ncclGetUniqueId. */
int32_t mxlo_rccl_unique_id(void *id_out);
/* every rank (collective): ncclCommInitRank on the CURRENT HIP device. */
int32_t mxlo_rccl_comm_create(int32_t rank, int32_t world, const void *id, void **comm_out);
int32_t mxlo_rccl_comm_destroy(void *comm);
/* an `mxlo_allreduce_fn`: user = the communicator; sums `count` doubles in place on `stream`. */
int32_t mxlo_rccl_allreduce_hook(void *user, void *dev_buf, int64_t count, void *stream);
const char *mxlo_rccl_last_error(void);
/* ncclCommAbort: tear the communicator down without waiting for outstanding collectives (a peer died, a collective
 * timed out). mxlo_rccl_comm_create itself refuses a communicator whose ncclCommCount / ncclCommUserRank differ from
 * what was asked for (a stale or foreign unique id). */
int32_t mxlo_rccl_comm_abort(void *comm);
/* What the communicator itself reports: ncclCommCount, ncclCommUserRank, ncclCommCuDevice and that device's PCI bus id
 * (hipDeviceGetPCIBusId; `pci_bus_id` may be NULL). A multi-rank bench line records these as its proof that N distinct
 * devices took part. */
int32_t mxlo_rccl_comm_info(void *comm, int32_t *ranks, int32_t *user_rank, int32_t *device, char *pci_bus_id, int32_t pci_len);
/* Preflight of a transport BEFORE any timed or production collective (every rank calls it; collective): for the three
 * payload sizes of the hot path — 8 B (opHouseholder), 320 B (forward L-BFGS m = 20), 6912 B (solve_shifted_system!
 * m = 20) — (1) an all-reduce with an exactly representable known answer (a missing / duplicated rank or a wrong sum
 * shows), (2) an all-reduce of inexact sums whose result bits are compared across ranks (all ranks must hold IDENTICAL
 * bits: they drive replicated control flow), (3) a verdict round so that every rank returns the same
 * status, (4) the latency of `reps` back-to-back all-reduces on `stream` (HIP events) -> latency_us[0..2] in microseconds.
 * Every wait is bounded by `timeout_ms` (hipStreamQuery + ncclCommGetAsyncError polling); on failure the error text
 * names the payload, the phase and this rank. `mxlo_rccl_preflight` tests the native hook on `comm`;
 * `mxlo_rccl_preflight_hook` tests ANY `mxlo_allreduce_fn` (the loopback and peer transports of the shard API, a host
 * runtime's own hook); `comm` there is only used for asynchronous-error polling and may be NULL. */
int32_t mxlo_rccl_preflight(void *comm, void *stream, int32_t reps, int32_t timeout_ms, double latency_us[3]);
int32_t mxlo_rccl_preflight_hook(int32_t (*hook)(void *, void *, int64_t, void *), void *user, void *comm, int32_t rank,
                                 int32_t world, void *stream, int32_t reps, int32_t timeout_ms, double latency_us[3]);

/* ======================================================================================================
 *  Single-process multi-device API ("host code stays in one Julia process").
 *
 *  The reference's device extension is a single-process model (ext/LinearOperatorsAMDGPUExt.jl:6); a Julia
 *  host has no torch.distributed and should not need MPI to drive the 8 GPUs of one node. One call builds,
 *  for every listed device, a non-blocking stream, an `mxlo_ctx` on it, an RCCL communicator
 *  (ncclCommInitAll) installed as that ctx's all-reduce hook, and one worker thread. A `_sharded` entry point
 *  takes PER-DEVICE POINTER ARRAYS (shard i = a contiguous row range of length n_local[i] living on device i)
 *  and runs the ordinary entry point of mxlo.h on every shard from that shard's worker thread; the only
 *  cross-device traffic is the scalar all-reduce inside the hook, enqueued on the shard's own stream
 *  (one thread per communicator: no ncclGroupStart/End choreography, and every kernel sequence of libmxlo works
 *  sharded unchanged). A call returns once every shard has ENQUEUED its work; `mxlo_shard_ctx_sync` waits.
 *  All shards end up with bit-identical replicated scalars (ys, Gram matrices, accept/reject of push!).
 *
 *  Listing one device id several times selects the LOOPBACK transport (several shards on one GPU, summed in
 *  fixed rank order through stream events): a test/debug mode for boxes with a single GPU.
 *  Status codes are those of mxlo.h; `mxlo_shard_last_error()` names the failing shard. */
typedef struct mxlo_ctx mxlo_ctx;
typedef struct mxlo_qn mxlo_qn;
typedef struct mxlo_shard_ctx mxlo_shard_ctx;
typedef struct mxlo_qn_sharded mxlo_qn_sharded;

/* dev_ids == NULL: devices 0 .. ndev-1. */
int32_t mxlo_shard_ctx_create(int32_t ndev, const int32_t *dev_ids, mxlo_shard_ctx **out);
/* ... with the transport of the scalar all-reduce chosen explicitly:
 *   MXLO_SHARD_AUTO     RCCL for distinct devices, loopback for repeated ids (what mxlo_shard_ctx_create does); the
 *                       environment variable MXLO_SHARD_TRANSPORT = rccl | loopback | peer overrides it;
 *   MXLO_SHARD_RCCL     ncclCommInitAll + ncclAllReduce on every shard's stream;
 *   MXLO_SHARD_LOOPBACK several shards on ONE device, summed in fixed rank order through stream events (debug);
 *   MXLO_SHARD_PEER     the peer-mapped ONE-SHOT exchange: per shard a mailbox in fine-grained device
 *                       memory that every other device stores into over xGMI (hipDeviceEnablePeerAccess; pinned host
 *                       memory when a pair has no peer access, or with MXLO_PEER_MEM=host), one small kernel per
 *                       collective (post, bounded poll of the own mailbox, sum in FIXED RANK ORDER: identical bits on every
 *                       shard), no RCCL call. Works with distinct devices and with repeated ids (one-GPU test shape). A
 *                       shard whose peers never post gives up after its timeout, stores NaN and the next call / sync
 *                       returns MXLO_EREDUCE (the ctx is unusable afterwards). */
#define MXLO_SHARD_AUTO 0
#define MXLO_SHARD_RCCL 1
#define MXLO_SHARD_LOOPBACK 2
#define MXLO_SHARD_PEER 3
int32_t mxlo_shard_ctx_create_ex(int32_t ndev, const int32_t *dev_ids, int32_t transport, mxlo_shard_ctx **out);
int32_t mxlo_shard_ctx_transport(mxlo_shard_ctx *s);
/* device ordinal and PCI bus id of shard i, and the number of ranks the transport itself reports (ncclCommCount) */
int32_t mxlo_shard_ctx_info(mxlo_shard_ctx *s, int32_t i, int32_t *device, int32_t *ranks_seen, char *pci_bus_id, int32_t pci_len);
/* mxlo_rccl_preflight_hook on every shard, through the hook the `_sharded` entry points use: known-answer sums, identical
 * bits on all shards, agreed verdict, and the latency (slowest shard) of the 8 B / 320 B / 6912 B all-reduce. */
int32_t mxlo_shard_ctx_preflight(mxlo_shard_ctx *s, int32_t reps, int32_t timeout_ms, double latency_us[3]);
/* test hooks of the peer transport: "peer_drop" (shard index that never posts; -1 none), "peer_timeout_ms" */
int32_t mxlo_shard_ctx_debug(mxlo_shard_ctx *s, const char *key, int64_t value);
int32_t mxlo_shard_ctx_destroy(mxlo_shard_ctx *s);
int32_t mxlo_shard_ctx_ndev(mxlo_shard_ctx *s);
int32_t mxlo_shard_ctx_device(mxlo_shard_ctx *s, int32_t i);
int32_t mxlo_shard_ctx_is_loopback(mxlo_shard_ctx *s);
/* the ctx of shard i: allocate / copy that shard's vectors with mxlo_malloc / mxlo_memcpy_* on it */
mxlo_ctx *mxlo_shard_ctx_get(mxlo_shard_ctx *s, int32_t i);
int32_t mxlo_shard_ctx_sync(mxlo_shard_ctx *s);
const char *mxlo_shard_last_error(void);

/* mulHouseholder! (src/linalg.jl:77-83), rows sharded: one 8-byte all-reduce of h'v between the two passes. */
int32_t mxlo_householder_mul_sharded(mxlo_shard_ctx *s, int32_t dtype, void *const *res, const void *const *h,
                                     const void *const *v, const int64_t *n_local, double alpha, double beta,
                                     int32_t flags);
/* mulSquareOpDiagonal! (src/special-operators.jl:125-131), rows sharded: independent shards, no exchange. */
int32_t mxlo_diag_mul_sharded(mxlo_shard_ctx *s, int32_t dtype, void *const *res, const void *const *d,
                              const void *const *v, const int64_t *n_local, double alpha, double beta, int32_t flags);

/* LBFGSData / LSR1Data (src/lbfgs.jl:26-57, src/lsr1.jl:19-34) with every stored vector row-sharded. */
int32_t mxlo_qn_create_sharded(mxlo_shard_ctx *s, int32_t kind, int32_t dtype, const int64_t *n_local, int64_t mem,
                               int32_t scaling, int32_t damped, double sigma2, double sigma3, mxlo_qn_sharded **out);
int32_t mxlo_qn_destroy_sharded(mxlo_qn_sharded *q);
mxlo_qn *mxlo_qn_sharded_get(mxlo_qn_sharded *q, int32_t i);
/* push!(op, s, y) (src/lbfgs.jl:269-287, src/lsr1.jl:119-184): 2-5 doubles all-reduced; `accepted` is replicated. */
int32_t mxlo_qn_push_sharded(mxlo_qn_sharded *q, const void *const *s, const void *const *y, int32_t *accepted);
/* lbfgs_multiply / lsr1_multiply (src/lbfgs.jl:117-154,173-202; src/lsr1.jl:89-107): one all-reduce of <= 2 mem doubles. */
int32_t mxlo_qn_mul_sharded(mxlo_qn_sharded *q, void *const *res, const void *const *x, double alpha, double beta,
                            int32_t flags);
int32_t mxlo_qn_mul_shifted_sharded(mxlo_qn_sharded *q, void *const *res, const void *const *x, double alpha,
                                    double beta, double sigma, int32_t flags);
/* solve_shifted_system! (src/utilities.jl:207-248), forward L-BFGS. */
int32_t mxlo_qn_solve_shifted_sharded(mxlo_qn_sharded *q, void *const *x, const void *const *b, double sigma);
int32_t mxlo_qn_diag_sharded(mxlo_qn_sharded *q, void *const *d);
int32_t mxlo_qn_reset_sharded(mxlo_qn_sharded *q);
/* replicated scalars as seen by shard i (layout of mxlo_qn_get_scalars) */
int32_t mxlo_qn_get_scalars_sharded(mxlo_qn_sharded *q, int32_t i, double scalars[5], double *ys, double *aux);

/* ======================================================================================================
 *  Peer transport for ONE PROCESS PER GPU (the shape bench.py and torch.distributed hosts run): the same one-shot
 *  exchange with the mailboxes in a POSIX shared-memory segment that every rank registers with its HIP runtime
 *  (hipHostRegister, mapped) — coherent by construction, no hipIpc handle exchange. Each collective is ONE kernel on the
 *  ctx stream; all ranks end up with identical bits (fixed rank order).
 *      rank 0 : mxlo_peer_comm_create_shm("/mxlo-<job token>", 0, world, 1, timeout_ms, &comm);   then a host barrier
 *      rank r : mxlo_peer_comm_create_shm("/mxlo-<job token>", r, world, 0, timeout_ms, &comm);
 *      every  : mxlo_ctx_set_allreduce(ctx, mxlo_peer_allreduce_hook, comm);
 *  mxlo_peer_comm_check (after a stream synchronisation) tells whether a gather of this rank timed out. */
typedef struct mxlo_peer_comm mxlo_peer_comm;
int32_t mxlo_peer_comm_create_shm(const char *name, int32_t rank, int32_t world, int32_t create, int32_t timeout_ms,
                                  mxlo_peer_comm **out);
int32_t mxlo_peer_comm_destroy(mxlo_peer_comm *comm);
int32_t mxlo_peer_allreduce_hook(void *user, void *dev_buf, int64_t count, void *stream);
int32_t mxlo_peer_comm_check(mxlo_peer_comm *comm);
int32_t mxlo_peer_comm_debug(mxlo_peer_comm *comm, const char *key, int64_t value);
const char *mxlo_peer_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
