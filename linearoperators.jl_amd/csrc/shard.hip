// shard.hip — libmxlo_rccl.so: ONE host process driving several GPUs (include/mxlo_rccl.h, "single-process API").
//
// The reference's device extension is a single-process model (ext/LinearOperatorsAMDGPUExt.jl:6): a Julia host has no
// torch.distributed and would need MPI just to exchange a communicator id. Here `mxlo_shard_ctx_create(ndev, ids)`
// sets everything up inside one process: per device a non-blocking stream, an mxlo_ctx on it, an RCCL communicator from
// ncclCommInitAll, the all-reduce hook of mxlo.h, and ONE WORKER THREAD. A `_sharded` call hands each worker the
// ordinary entry point for its row range; the hook inside it (the only cross-device traffic: a few scalars) is an
// ncclAllReduce on that device's stream — the standard one-thread-per-communicator pattern, so no ncclGroupStart/End
// choreography is needed and every libmxlo kernel sequence works sharded unchanged. Calls return when all devices
// have ENQUEUED their work (stream-ordered, like the unsharded ABI); push! synchronises, like the unsharded one.
//
// Loopback transport: when the same device id is listed more than once (several shards on one GPU — RCCL refuses
// that), the all-reduce is done by the shards' own streams with events: every shard sums all shards' scalars in
// FIXED RANK ORDER (bit-identical on every shard). It exists so the multi-shard logic can be exercised on a
// one-GPU box; it is not a performance path.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <initializer_list>
#include <utility>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mxlo.h"
#include "../../include/mxlo_rccl.h"

#define API extern "C" __attribute__((visibility("default")))

namespace {

thread_local char g_serr[512] = "";
void set_serr(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_serr, sizeof(g_serr), fmt, ap);
  va_end(ap);
}

constexpr int kMaxShards = 64;
constexpr int kLoopCap = 4096;   // doubles per loopback all-reduce (the hooks move <= 128 per call)

struct Worker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<int32_t()> job;
  bool has = false, done = false, quit = false;
  int32_t status = 0;
  std::string err;
};

struct Barrier {   // reusable host barrier for the loopback transport, with an abort path: a shard that fails
                   // anywhere (before or inside its collective) releases the others instead of leaving them waiting
  std::mutex mu;
  std::condition_variable cv;
  int n = 0, count = 0, gen = 0;
  bool aborted = false;
  bool wait() {   // false: the round was aborted (the caller returns an error)
    std::unique_lock<std::mutex> lk(mu);
    if (aborted) return false;
    const int g = gen;
    if (++count == n) {
      count = 0;
      ++gen;
      cv.notify_all();
      return true;
    }
    cv.wait(lk, [&] { return gen != g || aborted; });
    return !aborted;
  }
  void abort() {
    {
      std::lock_guard<std::mutex> lk(mu);
      aborted = true;
    }
    cv.notify_all();
  }
  void reset() {   // between calls (no worker is inside a collective: run_all holds the call mutex)
    std::lock_guard<std::mutex> lk(mu);
    aborted = false;
    count = 0;
  }
};

struct PtrPack {
  const double *p[kMaxShards];
};

__global__ void loop_sum_kernel(double *__restrict__ out, PtrPack in, int nshard, int64_t count) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  double s = in.p[0][i];
  for (int r = 1; r < nshard; ++r) s += in.p[r][i];   // fixed rank order: identical bits on every shard
  out[i] = s;
}

}  // namespace

struct mxlo_shard_ctx {
  int ndev = 0;
  bool loopback = false;
  std::vector<int> dev;
  std::vector<hipStream_t> streams;
  std::vector<mxlo_ctx *> ctx;
  std::vector<ncclComm_t> comms;
  std::vector<Worker *> workers;
  // loopback transport
  Barrier bar;
  std::vector<hipEvent_t> ev_ready, ev_read;
  std::vector<double *> tmp;
  std::vector<const double *> cur;
  struct HookUser {
    mxlo_shard_ctx *s;
    int rank;
  };
  std::vector<HookUser> users;
  std::mutex call_mu;       // one `_sharded` call at a time per shard ctx (two host threads may share one)
  bool poisoned = false;    // RCCL transport: a shard failed while its peers may already sit in a collective
};

struct mxlo_qn_sharded {
  mxlo_shard_ctx *s = nullptr;
  std::vector<mxlo_qn *> h;
  std::vector<int64_t> nloc;   // rows per shard (argument checks on the calling thread)
};

namespace {

int32_t rccl_hook(void *user, void *dev_buf, int64_t count, void *stream) {
  auto *u = (mxlo_shard_ctx::HookUser *)user;
  if (count <= 0) return 0;
  ncclResult_t r = ncclAllReduce(dev_buf, dev_buf, (size_t)count, ncclDouble, ncclSum, u->s->comms[u->rank],
                                 (hipStream_t)stream);
  return r == ncclSuccess ? 0 : 1;
}

int32_t loop_hook(void *user, void *dev_buf, int64_t count, void *stream_) {
  auto *u = (mxlo_shard_ctx::HookUser *)user;
  mxlo_shard_ctx *s = u->s;
  const int me = u->rank, n = s->ndev;
  hipStream_t st = (hipStream_t)stream_;
  if (count <= 0) return 0;
  auto fail = [&]() {   // release the peers: they return an error from their own wait()
    s->bar.abort();
    return 1;
  };
  if (count > kLoopCap) return fail();
  s->cur[me] = (const double *)dev_buf;
  if (hipEventRecord(s->ev_ready[me], st) != hipSuccess) return fail();
  if (!s->bar.wait()) return 1;                            // every shard's scalars are enqueued and recorded
  PtrPack pk;
  for (int r = 0; r < n; ++r) {
    pk.p[r] = s->cur[r];
    if (r != me && hipStreamWaitEvent(st, s->ev_ready[r], 0) != hipSuccess) return fail();
  }
  hipLaunchKernelGGL(loop_sum_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, s->tmp[me], pk, n, count);
  if (hipGetLastError() != hipSuccess) return fail();
  if (hipEventRecord(s->ev_read[me], st) != hipSuccess) return fail();
  if (!s->bar.wait()) return 1;                            // nobody overwrites its buffer before all have read it
  for (int r = 0; r < n; ++r)
    if (r != me && hipStreamWaitEvent(st, s->ev_read[r], 0) != hipSuccess) return fail();
  if (hipMemcpyAsync(dev_buf, s->tmp[me], sizeof(double) * count, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail();
  if (!s->bar.wait()) return 1;                            // events may be re-recorded by the next collective
  return 0;
}

void worker_main(Worker *w, int device) {
  (void)hipSetDevice(device);   // the entry points bind their ctx's device themselves; this keeps anything else the
                                // thread touches (RCCL's internal calls) on the right device from the first call on
  for (;;) {
    std::function<int32_t()> job;
    {
      std::unique_lock<std::mutex> lk(w->mu);
      w->cv.wait(lk, [&] { return w->has || w->quit; });
      if (w->quit) return;
      job = w->job;
      w->has = false;
    }
    const int32_t st = job();
    {
      std::lock_guard<std::mutex> lk(w->mu);
      w->status = st;
      w->err = st != 0 ? std::string(mxlo_last_error()) : std::string();
      w->done = true;
    }
    w->cv.notify_all();
  }
}

// run f(i) on worker i for every shard; returns the first non-zero status. Serialised per shard ctx. A shard that
// fails releases the loopback barrier (its peers return MXLO_EREDUCE instead of waiting forever); on the RCCL
// transport peers may already be inside a device-side collective that can never complete, so the ctx is poisoned:
// every later call returns MXLO_ESTATE and destroy aborts the communicators instead of draining them.
int32_t run_all(mxlo_shard_ctx *s, const std::function<int32_t(int)> &f) {
  std::lock_guard<std::mutex> call_lock(s->call_mu);
  if (s->poisoned) {
    set_serr("this shard ctx is unusable: an earlier sharded call failed on one device while the others were inside an "
             "RCCL collective; destroy it and create a new one");
    return MXLO_ESTATE;
  }
  if (s->loopback) s->bar.reset();
  for (int i = 0; i < s->ndev; ++i) {
    Worker *w = s->workers[i];
    {
      std::lock_guard<std::mutex> lk(w->mu);
      w->job = [&f, s, i]() {
        const int32_t st = f(i);
        if (st != 0 && s->loopback) s->bar.abort();
        return st;
      };
      w->done = false;
      w->has = true;
    }
    w->cv.notify_all();
  }
  int32_t first = 0;
  for (int i = 0; i < s->ndev; ++i) {
    Worker *w = s->workers[i];
    std::unique_lock<std::mutex> lk(w->mu);
    w->cv.wait(lk, [&] { return w->done; });
    if (w->status != 0 && first == 0) {
      first = w->status;
      set_serr("shard %d (device %d): %s", i, s->dev[i], w->err.c_str());
    }
  }
  if (first != 0 && !s->loopback && s->ndev > 1) s->poisoned = true;
  return first;
}

// Per-shard arguments are checked on the CALLING thread, before any worker starts: a bad pointer on one shard must not
// leave the other shards waiting in their collective.
int32_t check_shard_ptrs(const mxlo_shard_ctx *s, const char *fn, const int64_t *n_local,
                         std::initializer_list<std::pair<const char *, const void *const *>> arrays) {
  for (int i = 0; i < s->ndev; ++i) {
    if (n_local && n_local[i] < 0) {
      set_serr("%s: n_local[%d] = %lld is negative", fn, i, (long long)n_local[i]);
      return MXLO_EINVAL;
    }
    if (n_local && n_local[i] == 0) continue;   // an empty shard may pass NULL
    for (const auto &a : arrays)
      if (!a.second[i]) {
        set_serr("%s: %s[%d] is NULL", fn, a.first, i);
        return MXLO_EINVAL;
      }
  }
  return MXLO_OK;
}

}  // namespace

API const char *mxlo_shard_last_error(void) { return g_serr; }

API int32_t mxlo_shard_ctx_destroy(mxlo_shard_ctx *s) {
  if (!s) return MXLO_OK;
  for (Worker *w : s->workers) {
    if (!w) continue;
    {
      std::lock_guard<std::mutex> lk(w->mu);
      w->quit = true;
    }
    w->cv.notify_all();
    if (w->th.joinable()) w->th.join();
    delete w;
  }
  for (int i = 0; i < (int)s->ctx.size(); ++i) {
    (void)hipSetDevice(s->dev[i]);
    if (s->ctx[i]) (void)mxlo_ctx_destroy(s->ctx[i]);
    if (i < (int)s->comms.size() && s->comms[i]) (void)(s->poisoned ? ncclCommAbort(s->comms[i]) : ncclCommDestroy(s->comms[i]));
    if (i < (int)s->tmp.size() && s->tmp[i]) (void)hipFree(s->tmp[i]);
    if (i < (int)s->ev_ready.size() && s->ev_ready[i]) (void)hipEventDestroy(s->ev_ready[i]);
    if (i < (int)s->ev_read.size() && s->ev_read[i]) (void)hipEventDestroy(s->ev_read[i]);
    if (i < (int)s->streams.size() && s->streams[i]) (void)hipStreamDestroy(s->streams[i]);
  }
  delete s;
  return MXLO_OK;
}

API int32_t mxlo_shard_ctx_create(int32_t ndev, const int32_t *dev_ids, mxlo_shard_ctx **out) {
  if (!out || ndev < 1 || ndev > kMaxShards) {
    set_serr("mxlo_shard_ctx_create: ndev must be in 1..%d", kMaxShards);
    return MXLO_EINVAL;
  }
  int visible = 0;
  if (hipGetDeviceCount(&visible) != hipSuccess || visible < 1) {
    set_serr("mxlo_shard_ctx_create: no HIP device");
    return MXLO_EHIP;
  }
  int prev = 0;
  (void)hipGetDevice(&prev);
  mxlo_shard_ctx *s = new mxlo_shard_ctx();
  s->ndev = ndev;
  std::set<int> distinct;
  for (int i = 0; i < ndev; ++i) {
    const int d = dev_ids ? dev_ids[i] : i;        // NULL: devices 0 .. ndev-1
    if (d < 0 || d >= visible) {
      set_serr("mxlo_shard_ctx_create: device %d not in [0,%d)", d, visible);
      delete s;
      return MXLO_EINVAL;
    }
    s->dev.push_back(d);
    distinct.insert(d);
  }
  s->loopback = (int)distinct.size() < ndev;
  if (s->loopback && distinct.size() != 1) {
    set_serr("mxlo_shard_ctx_create: repeated device ids select the loopback transport, which needs ALL shards on one device");
    delete s;
    return MXLO_EINVAL;
  }
  s->streams.assign(ndev, nullptr);
  s->ctx.assign(ndev, nullptr);
  s->users.resize(ndev);
  int32_t st = MXLO_OK;
  for (int i = 0; i < ndev && st == MXLO_OK; ++i) {
    if (hipSetDevice(s->dev[i]) != hipSuccess || hipStreamCreateWithFlags(&s->streams[i], hipStreamNonBlocking) != hipSuccess) {
      set_serr("mxlo_shard_ctx_create: stream creation failed on device %d", s->dev[i]);
      st = MXLO_EHIP;
      break;
    }
    st = mxlo_ctx_create(s->dev[i], (void *)s->streams[i], &s->ctx[i]);
    if (st != MXLO_OK) set_serr("mxlo_shard_ctx_create: %s", mxlo_last_error());
    s->users[i] = {s, i};
  }
  if (st == MXLO_OK && !s->loopback) {
    s->comms.assign(ndev, nullptr);
    ncclResult_t r = ncclCommInitAll(s->comms.data(), ndev, s->dev.data());
    if (r != ncclSuccess) {
      set_serr("ncclCommInitAll: %s", ncclGetErrorString(r));
      st = MXLO_EREDUCE;
    }
  }
  if (st == MXLO_OK && s->loopback) {
    s->bar.n = ndev;
    s->ev_ready.assign(ndev, nullptr);
    s->ev_read.assign(ndev, nullptr);
    s->tmp.assign(ndev, nullptr);
    s->cur.assign(ndev, nullptr);
    (void)hipSetDevice(s->dev[0]);
    for (int i = 0; i < ndev && st == MXLO_OK; ++i) {
      if (hipEventCreateWithFlags(&s->ev_ready[i], hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&s->ev_read[i], hipEventDisableTiming) != hipSuccess ||
          hipMalloc((void **)&s->tmp[i], sizeof(double) * kLoopCap) != hipSuccess) {
        set_serr("mxlo_shard_ctx_create: loopback resources");
        st = MXLO_ENOMEM;
      }
    }
  }
  for (int i = 0; i < ndev && st == MXLO_OK; ++i)
    st = mxlo_ctx_set_allreduce(s->ctx[i], s->loopback ? loop_hook : rccl_hook, &s->users[i]);
  if (st == MXLO_OK) {
    for (int i = 0; i < ndev; ++i) {
      Worker *w = new Worker();
      s->workers.push_back(w);
      w->th = std::thread(worker_main, w, s->dev[i]);
    }
  }
  (void)hipSetDevice(prev);
  if (st != MXLO_OK) {
    mxlo_shard_ctx_destroy(s);
    return st;
  }
  *out = s;
  return MXLO_OK;
}

API int32_t mxlo_shard_ctx_ndev(mxlo_shard_ctx *s) { return s ? s->ndev : 0; }
API int32_t mxlo_shard_ctx_device(mxlo_shard_ctx *s, int32_t i) { return (s && i >= 0 && i < s->ndev) ? s->dev[i] : -1; }
API int32_t mxlo_shard_ctx_is_loopback(mxlo_shard_ctx *s) { return s && s->loopback ? 1 : 0; }
API mxlo_ctx *mxlo_shard_ctx_get(mxlo_shard_ctx *s, int32_t i) { return (s && i >= 0 && i < s->ndev) ? s->ctx[i] : nullptr; }

API int32_t mxlo_shard_ctx_sync(mxlo_shard_ctx *s) {
  if (!s) return MXLO_EINVAL;
  int32_t st = MXLO_OK;
  for (int i = 0; i < s->ndev; ++i) {
    const int32_t e = mxlo_ctx_sync(s->ctx[i]);
    if (e != MXLO_OK && st == MXLO_OK) st = e;
  }
  return st;
}

#define SHARD_REQUIRE(cond, ...)                                                                 \
  do {                                                                                           \
    if (!(cond)) {                                                                               \
      set_serr(__VA_ARGS__);                                                                     \
      return MXLO_EINVAL;                                                                        \
    }                                                                                            \
  } while (0)

// mulHouseholder! (src/linalg.jl:77-83) on row-sharded h, v, res: shard i holds rows of length n_local[i]; the one
// exchange is the 8-byte all-reduce of h'v between the dots pass and the update pass.
API int32_t mxlo_householder_mul_sharded(mxlo_shard_ctx *s, int32_t dtype, void *const *res, const void *const *h,
                                         const void *const *v, const int64_t *n_local, double alpha, double beta,
                                         int32_t flags) {
  SHARD_REQUIRE(s && res && h && v && n_local, "mxlo_householder_mul_sharded: NULL argument");
  if (int32_t e = check_shard_ptrs(s, "mxlo_householder_mul_sharded", n_local, {{"res", (const void *const *)res}, {"h", h}, {"v", v}})) return e;
  return run_all(s, [&](int i) {
    return mxlo_householder_mul(s->ctx[i], dtype, res[i], h[i], v[i], n_local[i], alpha, beta, flags);
  });
}

// mulSquareOpDiagonal! on row shards: independent, no exchange (same entry point per shard, for symmetry).
API int32_t mxlo_diag_mul_sharded(mxlo_shard_ctx *s, int32_t dtype, void *const *res, const void *const *d,
                                  const void *const *v, const int64_t *n_local, double alpha, double beta, int32_t flags) {
  SHARD_REQUIRE(s && res && d && v && n_local, "mxlo_diag_mul_sharded: NULL argument");
  if (int32_t e = check_shard_ptrs(s, "mxlo_diag_mul_sharded", n_local, {{"res", (const void *const *)res}, {"d", d}, {"v", v}})) return e;
  return run_all(s, [&](int i) {
    return mxlo_diag_mul(s->ctx[i], dtype, res[i], d[i], v[i], n_local[i], n_local[i], alpha, beta, flags);
  });
}

API int32_t mxlo_qn_destroy_sharded(mxlo_qn_sharded *q) {
  if (!q) return MXLO_OK;
  for (int i = 0; i < (int)q->h.size(); ++i)
    if (q->h[i]) (void)mxlo_qn_destroy(q->h[i]);
  delete q;
  return MXLO_OK;
}

// LBFGSData / LSR1Data with the panels row-sharded: shard i owns n_local[i] rows of every stored vector; the small
// scalar state (ys, Gram matrices, coefficients) is replicated and stays bit-identical through the all-reduce.
API int32_t mxlo_qn_create_sharded(mxlo_shard_ctx *s, int32_t kind, int32_t dtype, const int64_t *n_local, int64_t mem,
                                   int32_t scaling, int32_t damped, double sigma2, double sigma3, mxlo_qn_sharded **out) {
  SHARD_REQUIRE(s && n_local && out, "mxlo_qn_create_sharded: NULL argument");
  if (int32_t e = check_shard_ptrs(s, "mxlo_qn_create_sharded", n_local, {})) return e;
  mxlo_qn_sharded *q = new mxlo_qn_sharded();
  q->s = s;
  q->h.assign(s->ndev, nullptr);
  q->nloc.assign(n_local, n_local + s->ndev);
  const int32_t st = run_all(s, [&](int i) {
    return mxlo_qn_create(s->ctx[i], kind, dtype, n_local[i], mem, scaling, damped, sigma2, sigma3, &q->h[i]);
  });
  if (st != MXLO_OK) {
    mxlo_qn_destroy_sharded(q);
    return st;
  }
  *out = q;
  return MXLO_OK;
}

API mxlo_qn *mxlo_qn_sharded_get(mxlo_qn_sharded *q, int32_t i) {
  return (q && i >= 0 && i < (int)q->h.size()) ? q->h[i] : nullptr;
}

// push!(op, s, y): `accepted` is the replicated decision (identical on every shard by construction; checked).
API int32_t mxlo_qn_push_sharded(mxlo_qn_sharded *q, const void *const *sv, const void *const *yv, int32_t *accepted) {
  SHARD_REQUIRE(q && sv && yv && accepted, "mxlo_qn_push_sharded: NULL argument");
  if (int32_t e = check_shard_ptrs(q->s, "mxlo_qn_push_sharded", q->nloc.data(), {{"s", sv}, {"y", yv}})) return e;
  std::vector<int32_t> acc(q->h.size(), -1);
  const int32_t st = run_all(q->s, [&](int i) { return mxlo_qn_push(q->h[i], sv[i], yv[i], &acc[i]); });
  if (st != MXLO_OK) return st;
  for (size_t i = 1; i < acc.size(); ++i)
    if (acc[i] != acc[0]) {
      set_serr("mxlo_qn_push_sharded: shards disagree on accept/reject (%d vs %d): the all-reduce did not deliver "
               "identical scalars", acc[0], acc[i]);
      return MXLO_EREDUCE;
    }
  *accepted = acc[0];
  return MXLO_OK;
}

API int32_t mxlo_qn_mul_sharded(mxlo_qn_sharded *q, void *const *res, const void *const *x, double alpha, double beta,
                                int32_t flags) {
  SHARD_REQUIRE(q && res && x, "mxlo_qn_mul_sharded: NULL argument");
  if (int32_t e = check_shard_ptrs(q->s, "mxlo_qn_mul_sharded", q->nloc.data(), {{"res", (const void *const *)res}, {"x", x}})) return e;
  return run_all(q->s, [&](int i) { return mxlo_qn_mul(q->h[i], res[i], x[i], alpha, beta, flags); });
}

API int32_t mxlo_qn_mul_shifted_sharded(mxlo_qn_sharded *q, void *const *res, const void *const *x, double alpha,
                                        double beta, double sigma, int32_t flags) {
  SHARD_REQUIRE(q && res && x, "mxlo_qn_mul_shifted_sharded: NULL argument");
  if (int32_t e = check_shard_ptrs(q->s, "mxlo_qn_mul_shifted_sharded", q->nloc.data(), {{"res", (const void *const *)res}, {"x", x}})) return e;
  return run_all(q->s, [&](int i) { return mxlo_qn_mul_shifted(q->h[i], res[i], x[i], alpha, beta, sigma, flags); });
}

API int32_t mxlo_qn_solve_shifted_sharded(mxlo_qn_sharded *q, void *const *x, const void *const *b, double sigma) {
  SHARD_REQUIRE(q && x && b, "mxlo_qn_solve_shifted_sharded: NULL argument");
  if (int32_t e = check_shard_ptrs(q->s, "mxlo_qn_solve_shifted_sharded", q->nloc.data(), {{"x", (const void *const *)x}, {"b", b}})) return e;
  return run_all(q->s, [&](int i) { return mxlo_qn_solve_shifted(q->h[i], x[i], b[i], sigma); });
}

API int32_t mxlo_qn_diag_sharded(mxlo_qn_sharded *q, void *const *d) {
  SHARD_REQUIRE(q && d, "mxlo_qn_diag_sharded: NULL argument");
  if (int32_t e = check_shard_ptrs(q->s, "mxlo_qn_diag_sharded", q->nloc.data(), {{"d", (const void *const *)d}})) return e;
  return run_all(q->s, [&](int i) { return mxlo_qn_diag(q->h[i], d[i]); });
}

API int32_t mxlo_qn_reset_sharded(mxlo_qn_sharded *q) {
  SHARD_REQUIRE(q, "mxlo_qn_reset_sharded: NULL argument");
  return run_all(q->s, [&](int i) { return mxlo_qn_reset(q->h[i]); });
}

// the replicated scalars of shard `i` (all shards hold the same values; `i` lets a test check exactly that)
API int32_t mxlo_qn_get_scalars_sharded(mxlo_qn_sharded *q, int32_t i, double scalars[5], double *ys, double *aux) {
  SHARD_REQUIRE(q && i >= 0 && i < (int)q->h.size(), "mxlo_qn_get_scalars_sharded: bad shard index");
  int32_t st = MXLO_OK;
  const int32_t r = run_all(q->s, [&](int k) {
    // the opnorm bound of L-SR1 may launch lazily computed norms with their own all-reduce: every shard takes part
    double sc[5];
    std::vector<double> y(64 + 4096), a(64 + 4096);
    const int32_t e = mxlo_qn_get_scalars(q->h[k], k == i ? scalars : sc, k == i ? ys : (ys ? y.data() : nullptr),
                                          k == i ? aux : (aux ? a.data() : nullptr));
    if (k == i) st = e;
    return e;
  });
  return r != MXLO_OK ? r : st;
}
