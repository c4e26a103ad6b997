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

#ifdef __cplusplus
}
#endif
#endif
