// rccl_hook.cpp — libmxlo_rccl.so: the RCCL transport of the all-reduce hook (include/mxlo_rccl.h).
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/mxlo_rccl.h"

#define API extern "C" __attribute__((visibility("default")))

static thread_local char g_err[512] = "";
static int32_t fail(const char *what, ncclResult_t r) {
  snprintf(g_err, sizeof(g_err), "%s: %s", what, ncclGetErrorString(r));
  return 1;
}
static int32_t fail_msg(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
static int32_t fail_msg(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return 1;
}

API const char *mxlo_rccl_last_error(void) { return g_err; }

API int32_t mxlo_rccl_unique_id(void *id_out) {
  static_assert(sizeof(ncclUniqueId) <= MXLO_RCCL_ID_BYTES, "ncclUniqueId larger than MXLO_RCCL_ID_BYTES");
  if (!id_out) return fail_msg("mxlo_rccl_unique_id: id_out is NULL");
  ncclUniqueId id;
  ncclResult_t r = ncclGetUniqueId(&id);
  if (r != ncclSuccess) return fail("ncclGetUniqueId", r);
  memset(id_out, 0, MXLO_RCCL_ID_BYTES);
  memcpy(id_out, &id, sizeof(id));
  return 0;
}

API int32_t mxlo_rccl_comm_create(int32_t rank, int32_t world, const void *id_in, void **comm_out) {
  if (!id_in || !comm_out) return fail_msg("mxlo_rccl_comm_create: NULL argument");
  if (world < 1 || rank < 0 || rank >= world) return fail_msg("mxlo_rccl_comm_create: rank %d not in [0, %d)", rank, world);
  *comm_out = nullptr;
  ncclUniqueId id;
  memcpy(&id, id_in, sizeof(id));
  ncclComm_t comm = nullptr;
  ncclResult_t r = ncclCommInitRank(&comm, world, id, rank);
  if (r != ncclSuccess) return fail("ncclCommInitRank", r);
  // the communicator must describe what the caller asked for: a mismatch means a stale or foreign unique id
  int cnt = -1, ur = -1;
  if (ncclCommCount(comm, &cnt) != ncclSuccess || ncclCommUserRank(comm, &ur) != ncclSuccess || cnt != world || ur != rank) {
    (void)ncclCommAbort(comm);
    return fail_msg("mxlo_rccl_comm_create: communicator reports %d ranks / rank %d, asked for %d / %d", cnt, ur, world, rank);
  }
  *comm_out = (void *)comm;
  return 0;
}

API int32_t mxlo_rccl_comm_destroy(void *comm) {
  if (!comm) return 0;
  ncclResult_t r = ncclCommDestroy((ncclComm_t)comm);
  return r == ncclSuccess ? 0 : fail("ncclCommDestroy", r);
}

// Tear a communicator down WITHOUT waiting for outstanding collectives (a peer is gone, a collective timed out).
API int32_t mxlo_rccl_comm_abort(void *comm) {
  if (!comm) return 0;
  ncclResult_t r = ncclCommAbort((ncclComm_t)comm);
  return r == ncclSuccess ? 0 : fail("ncclCommAbort", r);
}

API int32_t mxlo_rccl_comm_info(void *comm, int32_t *ranks, int32_t *user_rank, int32_t *device, char *pci_bus_id,
                                int32_t pci_len) {
  if (!comm) return fail_msg("mxlo_rccl_comm_info: comm is NULL");
  int cnt = -1, ur = -1, dev = -1;
  ncclResult_t r;
  if ((r = ncclCommCount((ncclComm_t)comm, &cnt)) != ncclSuccess) return fail("ncclCommCount", r);
  if ((r = ncclCommUserRank((ncclComm_t)comm, &ur)) != ncclSuccess) return fail("ncclCommUserRank", r);
  if ((r = ncclCommCuDevice((ncclComm_t)comm, &dev)) != ncclSuccess) return fail("ncclCommCuDevice", r);
  if (ranks) *ranks = cnt;
  if (user_rank) *user_rank = ur;
  if (device) *device = dev;
  if (pci_bus_id && pci_len > 0) {
    pci_bus_id[0] = 0;
    if (hipDeviceGetPCIBusId(pci_bus_id, pci_len, dev) != hipSuccess) {
      (void)hipGetLastError();
      snprintf(pci_bus_id, (size_t)pci_len, "?");
    }
  }
  return 0;
}

API int32_t mxlo_rccl_allreduce_hook(void *user, void *dev_buf, int64_t count, void *stream) {
  if (!user || !dev_buf || count < 0) return 1;
  if (count == 0) return 0;
  ncclResult_t r = ncclAllReduce(dev_buf, dev_buf, (size_t)count, ncclDouble, ncclSum, (ncclComm_t)user,
                                 (hipStream_t)stream);
  return r == ncclSuccess ? 0 : fail("ncclAllReduce", r);
}

// ---- preflight: prove, through an `mxlo_allreduce_fn` hook, that every rank is there and the sums are right -------------
// Bounded wait on a stream: polls hipStreamQuery (and the communicator's asynchronous error state when there is one)
// until the stream drains or `timeout_ms` passes. 0 = drained, 1 = error (g_err), 2 = timed out.
static int wait_stream(hipStream_t st, ncclComm_t comm, int timeout_ms, const char *phase) {
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t q = hipStreamQuery(st);
    if (q == hipSuccess) return 0;
    if (q != hipErrorNotReady) {
      fail_msg("%s: %s", phase, hipGetErrorString(q));
      return 1;
    }
    if (comm) {
      ncclResult_t ar = ncclSuccess;
      if (ncclCommGetAsyncError(comm, &ar) == ncclSuccess && ar != ncclSuccess && ar != ncclInProgress) {
        fail("asynchronous communicator error", ar);
        return 1;
      }
    }
    const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
    if (ms > timeout_ms) {
      fail_msg("%s did not complete within %d ms (a rank is missing from the collective, or the fabric is stuck)", phase, timeout_ms);
      return 2;
    }
    std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
}

// The three payloads of the hot path: 8 B (Householder), 320 B (forward L-BFGS m = 20), 6.9 KB (shifted solve
// m = 20: 864 doubles). Per payload:
//   1. known answer: every rank contributes (rank + 1) * (1 + i / 1024) in element i — exactly representable, so the
//      sum world (world + 1) / 2 * (1 + i / 1024) is exact whatever the reduction order: a rank that is missing, counted
//      twice, or summed wrongly shows;
//   2. identical bits: every rank contributes pseudo-random doubles (inexact sums — the order matters); each rank then
//      splits the bits of ITS result into four 16-bit fields, the fields are all-reduced (exact: < 2^22), and a rank whose
//      fields differ from field_sum / world holds a result some other rank does not;
//   3. verdict: the number of ranks that failed 1 or 2 is all-reduced, so that EVERY rank returns the same status (one
//      rank erroring out alone would leave the others waiting in their next collective);
//   4. latency: `reps` back-to-back in-place all-reduces on `stream` between two HIP events.
//   hook / user : the transport under test (mxlo_rccl_allreduce_hook + communicator, or any mxlo_allreduce_fn)
//   comm        : the communicator for asynchronous-error polling (may be NULL)
// Returns 0 and fills latency_us[3]; non-zero with mxlo_rccl_last_error() naming the phase otherwise. Every wait is
// bounded by `timeout_ms`.
API int32_t mxlo_rccl_preflight_hook(int32_t (*hook)(void *, void *, int64_t, void *), void *user, void *comm, int32_t rank,
                                     int32_t world, void *stream, int32_t reps, int32_t timeout_ms, double latency_us[3]) {
  if (!hook || world < 1 || world > 64 || rank < 0 || rank >= world || reps < 1 || !latency_us)
    return fail_msg("mxlo_rccl_preflight_hook: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int64_t counts[3] = {1, 40, 864};
  const int64_t kMax = 864 * 4;
  double *dev = nullptr;
  if (hipMalloc((void **)&dev, sizeof(double) * kMax) != hipSuccess) return fail_msg("preflight: hipMalloc failed");
  hipEvent_t e0 = nullptr, e1 = nullptr;
  std::vector<double> host(kMax), mine(kMax);
  auto done = [&](int32_t code) {
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(dev);
    return code;
  };
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return done(fail_msg("preflight: hipEventCreate failed"));
  char phase[128], why[256];
  // one in-place all-reduce of host[0..cnt) through the hook, result back in host; 0 ok, non-zero: g_err set
  auto exchange = [&](int64_t cnt) -> int {
    if (hipMemcpyAsync(dev, host.data(), sizeof(double) * cnt, hipMemcpyHostToDevice, st) != hipSuccess) return fail_msg("%s: H2D copy failed", phase);
    if (hook(user, dev, cnt, st) != 0) {
      char inner[256];
      snprintf(inner, sizeof(inner), "%s", g_err);
      return fail_msg("%s: the all-reduce hook returned an error (%s)", phase, inner);
    }
    if (hipMemcpyAsync(host.data(), dev, sizeof(double) * cnt, hipMemcpyDeviceToHost, st) != hipSuccess) return fail_msg("%s: D2H copy failed", phase);
    return wait_stream(st, (ncclComm_t)comm, timeout_ms, phase);
  };
  for (int k = 0; k < 3; ++k) {
    const int64_t cnt = counts[k];
    int bad = 0;
    why[0] = 0;
    // 1. known answer
    snprintf(phase, sizeof(phase), "preflight %lld B known-answer all-reduce (rank %d of %d)", (long long)(cnt * 8), rank, world);
    for (int64_t i = 0; i < cnt; ++i) host[i] = (double)(rank + 1) * (1.0 + (double)i / 1024.0);
    if (exchange(cnt) != 0) return done(1);
    const double tri = 0.5 * (double)world * (double)(world + 1);
    for (int64_t i = 0; i < cnt && !bad; ++i) {
      const double want = tri * (1.0 + (double)i / 1024.0);
      if (host[i] != want) {
        bad = 1;
        snprintf(why, sizeof(why), "element %lld is %.17g, expected %.17g (%s)", (long long)i, host[i], want,
                 host[i] == (double)(rank + 1) * (1.0 + (double)i / 1024.0) ? "no other rank contributed" : "wrong sum");
      }
    }
    // 2. identical bits on inexact sums
    snprintf(phase, sizeof(phase), "preflight %lld B identical-bits all-reduce (rank %d of %d)", (long long)(cnt * 8), rank, world);
    for (int64_t i = 0; i < cnt; ++i) {
      uint64_t z = (uint64_t)(rank + 1) * 0x9E3779B97F4A7C15ull + (uint64_t)i * 0xBF58476D1CE4E5B9ull + (uint64_t)k;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
      z ^= z >> 31;
      host[i] = (double)(z >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0;
    }
    if (exchange(cnt) != 0) return done(1);
    for (int64_t i = 0; i < cnt; ++i) {
      uint64_t b;
      memcpy(&b, &host[i], 8);
      for (int j = 0; j < 4; ++j) mine[4 * i + j] = (double)((b >> (16 * j)) & 0xffffu);
    }
    for (int64_t i = 0; i < 4 * cnt; ++i) host[i] = mine[i];
    if (exchange(4 * cnt) != 0) return done(1);
    for (int64_t i = 0; i < 4 * cnt && !bad; ++i)
      if (host[i] != (double)world * mine[i]) {
        bad = 1;
        snprintf(why, sizeof(why), "the reduced value of element %lld differs between ranks (bit field %lld)", (long long)(i / 4), (long long)(i % 4));
      }
    // 3. verdict, agreed by all ranks
    snprintf(phase, sizeof(phase), "preflight %lld B verdict round (rank %d of %d)", (long long)(cnt * 8), rank, world);
    host[0] = (double)bad;
    if (exchange(1) != 0) return done(1);
    if (host[0] != 0.0)
      return done(fail_msg("preflight all-reduce of %lld B FAILED on %d of %d rank(s)%s%s", (long long)(cnt * 8), (int)host[0], world,
                           bad ? "; this rank: " : " (not this one)", bad ? why : ""));
    // 4. latency
    snprintf(phase, sizeof(phase), "preflight %lld B latency loop (rank %d of %d)", (long long)(cnt * 8), rank, world);
    for (int w = 0; w < 5; ++w)
      if (hook(user, dev, cnt, st) != 0) return done(fail_msg("%s: the hook returned an error", phase));
    if (hipEventRecord(e0, st) != hipSuccess) return done(fail_msg("%s: hipEventRecord failed", phase));
    for (int r = 0; r < reps; ++r)
      if (hook(user, dev, cnt, st) != 0) return done(fail_msg("%s: the hook returned an error", phase));
    if (hipEventRecord(e1, st) != hipSuccess) return done(fail_msg("%s: hipEventRecord failed", phase));
    if (wait_stream(st, (ncclComm_t)comm, timeout_ms, phase) != 0) return done(1);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return done(fail_msg("%s: hipEventElapsedTime failed", phase));
    latency_us[k] = (double)ms * 1e3 / reps;
  }
  return done(0);
}

API int32_t mxlo_rccl_preflight(void *comm, void *stream, int32_t reps, int32_t timeout_ms, double latency_us[3]) {
  if (!comm) return fail_msg("mxlo_rccl_preflight: comm is NULL");
  int cnt = 0, ur = 0;
  ncclResult_t r;
  if ((r = ncclCommCount((ncclComm_t)comm, &cnt)) != ncclSuccess) return fail("ncclCommCount", r);
  if ((r = ncclCommUserRank((ncclComm_t)comm, &ur)) != ncclSuccess) return fail("ncclCommUserRank", r);
  return mxlo_rccl_preflight_hook(mxlo_rccl_allreduce_hook, comm, comm, ur, cnt, stream, reps, timeout_ms, latency_us);
}
