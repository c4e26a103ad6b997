// rccl_hook.cpp — libmxlo_rccl.so: the RCCL transport of the all-reduce hook (include/mxlo_rccl.h).
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>

#include "../../include/mxlo_rccl.h"

#define API extern "C" __attribute__((visibility("default")))

static thread_local char g_err[256] = "";
static int32_t fail(const char *what, ncclResult_t r) {
  snprintf(g_err, sizeof(g_err), "%s: %s", what, ncclGetErrorString(r));
  return 1;
}

API const char *mxlo_rccl_last_error(void) { return g_err; }

API int32_t mxlo_rccl_unique_id(void *id_out) {
  static_assert(sizeof(ncclUniqueId) <= MXLO_RCCL_ID_BYTES, "ncclUniqueId larger than MXLO_RCCL_ID_BYTES");
  if (!id_out) return 1;
  ncclUniqueId id;
  ncclResult_t r = ncclGetUniqueId(&id);
  if (r != ncclSuccess) return fail("ncclGetUniqueId", r);
  memset(id_out, 0, MXLO_RCCL_ID_BYTES);
  memcpy(id_out, &id, sizeof(id));
  return 0;
}

API int32_t mxlo_rccl_comm_create(int32_t rank, int32_t world, const void *id_in, void **comm_out) {
  if (!id_in || !comm_out || world < 1 || rank < 0 || rank >= world) return 1;
  ncclUniqueId id;
  memcpy(&id, id_in, sizeof(id));
  ncclComm_t comm = nullptr;
  ncclResult_t r = ncclCommInitRank(&comm, world, id, rank);
  if (r != ncclSuccess) return fail("ncclCommInitRank", r);
  *comm_out = (void *)comm;
  return 0;
}

API int32_t mxlo_rccl_comm_destroy(void *comm) {
  if (!comm) return 0;
  ncclResult_t r = ncclCommDestroy((ncclComm_t)comm);
  return r == ncclSuccess ? 0 : fail("ncclCommDestroy", r);
}

API int32_t mxlo_rccl_allreduce_hook(void *user, void *dev_buf, int64_t count, void *stream) {
  if (!user || !dev_buf || count < 0) return 1;
  if (count == 0) return 0;
  ncclResult_t r = ncclAllReduce(dev_buf, dev_buf, (size_t)count, ncclDouble, ncclSum, (ncclComm_t)user,
                                 (hipStream_t)stream);
  return r == ncclSuccess ? 0 : fail("ncclAllReduce", r);
}
