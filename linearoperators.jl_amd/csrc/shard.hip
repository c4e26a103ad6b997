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
//
// Peer transport (MXLO_SHARD_PEER, round 5; peer.hip): per shard a mailbox every other shard's device can store into
// (fine-grained device memory + peer access; pinned host memory as the fallback) and ONE small kernel per collective —
// post into every mailbox, poll the own one, add in fixed rank order. No RCCL call. With all shards on ONE device (the
// test shape of a one-GPU box) the kernel is issued as its two halves with a host barrier between them, so that every
// post is enqueued ahead of every polling gather whatever hardware queue the streams share.
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
#include "peer.h"

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
  bool poisoned = false;    // RCCL / peer transport: a shard failed while its peers may already sit in a collective
  int transport = MXLO_SHARD_RCCL;
  bool same_device = false; // every shard on one device (loopback, or the peer transport's one-GPU test shape)
  bool uses_bar = false;    // the transport synchronises its worker threads with `bar`
  mxlo_allreduce_fn hook_fn = nullptr;
  // peer transport
  std::vector<mxlo_peer::Comm> pc;
  std::vector<void *> mailbox;
  bool mailbox_on_host = false;
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

// Peer transport: one kernel per collective (distinct devices); on ONE device post and gather are two launches with a
// host barrier between them (see the header comment). The sequence number advances identically on every shard.
int32_t peer_hook(void *user, void *dev_buf, int64_t count, void *stream_) {
  auto *u = (mxlo_shard_ctx::HookUser *)user;
  mxlo_shard_ctx *s = u->s;
  mxlo_peer::Comm &c = s->pc[u->rank];
  hipStream_t st = (hipStream_t)stream_;
  if (count <= 0) return 0;
  auto fail = [&]() {
    if (s->uses_bar) s->bar.abort();
    return 1;
  };
  if (c.dead || mxlo_peer::comm_fault(&c) != 0) return fail();
  ++c.seq;
  if (!s->same_device) return mxlo_peer::comm_launch(&c, (double *)dev_buf, count, st, 3) == 0 ? 0 : fail();
  if (mxlo_peer::comm_launch(&c, (double *)dev_buf, count, st, 1) != 0) return fail();
  if (!s->bar.wait()) return 1;                            // every shard's post is enqueued
  if (mxlo_peer::comm_launch(&c, (double *)dev_buf, count, st, 2) != 0) return fail();
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
  if (s->uses_bar) s->bar.reset();
  for (int i = 0; i < s->ndev; ++i) {
    Worker *w = s->workers[i];
    {
      std::lock_guard<std::mutex> lk(w->mu);
      w->job = [&f, s, i]() {
        const int32_t st = f(i);
        if (st != 0 && s->uses_bar) s->bar.abort();
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
  // a shard that failed while its peers may sit in a device-side collective (RCCL, or a polling peer gather that will
  // run into its timeout): nothing on this ctx can be trusted to line up again
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
    if (i < (int)s->pc.size()) mxlo_peer::comm_release_common(&s->pc[i]);
    if (i < (int)s->mailbox.size() && s->mailbox[i]) (void)(s->mailbox_on_host ? hipHostFree(s->mailbox[i]) : hipFree(s->mailbox[i]));
    if (i < (int)s->ev_ready.size() && s->ev_ready[i]) (void)hipEventDestroy(s->ev_ready[i]);
    if (i < (int)s->ev_read.size() && s->ev_read[i]) (void)hipEventDestroy(s->ev_read[i]);
    if (i < (int)s->streams.size() && s->streams[i]) (void)hipStreamDestroy(s->streams[i]);
  }
  delete s;
  return MXLO_OK;
}

namespace {
// mailboxes of the peer transport: fine-grained device memory with peer access between every pair of devices; pinned
// (portable, mapped) host memory when a pair cannot reach each other or MXLO_PEER_MEM=host asks for it
int32_t peer_setup(mxlo_shard_ctx *s) {
  const int n = s->ndev;
  const size_t bytes = mxlo_peer::mailbox_words(n) * sizeof(unsigned long long);
  s->pc.assign(n, mxlo_peer::Comm());
  s->mailbox.assign(n, nullptr);
  const char *mem = getenv("MXLO_PEER_MEM");
  bool host = mem && !strcmp(mem, "host");
  if (!host && !s->same_device) {
    for (int i = 0; i < n && !host; ++i)
      for (int j = 0; j < n && !host; ++j) {
        int can = 0;
        if (i != j && (hipDeviceCanAccessPeer(&can, s->dev[i], s->dev[j]) != hipSuccess || !can)) host = true;
      }
  }
  if (!host) {
    for (int i = 0; i < n; ++i) {
      if (hipSetDevice(s->dev[i]) != hipSuccess) return MXLO_EHIP;
      void *p = nullptr;
      if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        if (s->same_device && hipMalloc(&p, bytes) == hipSuccess) {
          // one device: ordinary device memory is coherent for the system-scope accesses of the exchange kernel
        } else {
          (void)hipGetLastError();
          host = true;
          break;
        }
      }
      s->mailbox[i] = p;
      if (hipMemset(p, 0, bytes) != hipSuccess) return MXLO_EHIP;
    }
    if (host) {
      for (int i = 0; i < n; ++i)
        if (s->mailbox[i]) {
          (void)hipFree(s->mailbox[i]);
          s->mailbox[i] = nullptr;
        }
    }
  }
  if (!host && !s->same_device) {
    for (int i = 0; i < n; ++i) {
      if (hipSetDevice(s->dev[i]) != hipSuccess) return MXLO_EHIP;
      for (int j = 0; j < n; ++j) {
        if (i == j) continue;
        const hipError_t e = hipDeviceEnablePeerAccess(s->dev[j], 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
          set_serr("mxlo_shard_ctx_create: hipDeviceEnablePeerAccess(%d -> %d): %s", s->dev[i], s->dev[j], hipGetErrorString(e));
          return MXLO_EHIP;
        }
        (void)hipGetLastError();
      }
    }
  }
  if (host) {
    s->mailbox_on_host = true;
    for (int i = 0; i < n; ++i) {
      void *p = nullptr;
      if (hipHostMalloc(&p, bytes, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) {
        set_serr("mxlo_shard_ctx_create: pinned host memory for the peer mailboxes");
        return MXLO_ENOMEM;
      }
      memset(p, 0, bytes);
      s->mailbox[i] = p;
    }
  }
  for (int i = 0; i < n; ++i) {
    if (hipSetDevice(s->dev[i]) != hipSuccess) return MXLO_EHIP;
    if (mxlo_peer::comm_init_common(&s->pc[i], i, n, 0) != 0) {
      set_serr("mxlo_shard_ctx_create: %s", mxlo_peer::last_error());
      return MXLO_EHIP;
    }
    for (int r = 0; r < n; ++r) {
      void *dp = s->mailbox[r];
      if (host && hipHostGetDevicePointer(&dp, s->mailbox[r], 0) != hipSuccess) return MXLO_EHIP;
      s->pc[i].mb[r] = (unsigned long long *)dp;
    }
  }
  return MXLO_OK;
}
}  // namespace

// transport: MXLO_SHARD_AUTO (RCCL for distinct devices, loopback for repeated ids; the environment variable
// MXLO_SHARD_TRANSPORT = rccl | loopback | peer overrides), MXLO_SHARD_RCCL, MXLO_SHARD_LOOPBACK, MXLO_SHARD_PEER.
API int32_t mxlo_shard_ctx_create_ex(int32_t ndev, const int32_t *dev_ids, int32_t transport, mxlo_shard_ctx **out) {
  if (!out || ndev < 1 || ndev > kMaxShards) {
    set_serr("mxlo_shard_ctx_create: ndev must be in 1..%d", kMaxShards);
    return MXLO_EINVAL;
  }
  *out = nullptr;
  if (transport == MXLO_SHARD_AUTO) {
    const char *e = getenv("MXLO_SHARD_TRANSPORT");
    if (e && !strcmp(e, "rccl")) transport = MXLO_SHARD_RCCL;
    else if (e && !strcmp(e, "loopback")) transport = MXLO_SHARD_LOOPBACK;
    else if (e && !strcmp(e, "peer")) transport = MXLO_SHARD_PEER;
    else if (e && e[0]) {
      set_serr("mxlo_shard_ctx_create: MXLO_SHARD_TRANSPORT='%s' is not one of rccl, loopback, peer", e);
      return MXLO_EINVAL;
    }
  }
  if (transport < MXLO_SHARD_AUTO || transport > MXLO_SHARD_PEER) {
    set_serr("mxlo_shard_ctx_create_ex: unknown transport %d", transport);
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
  const bool repeated = (int)distinct.size() < ndev;
  if (repeated && distinct.size() != 1) {
    set_serr("mxlo_shard_ctx_create: repeated device ids put several shards on one GPU (loopback / peer test shape), which needs ALL shards on one device");
    delete s;
    return MXLO_EINVAL;
  }
  if (transport == MXLO_SHARD_AUTO) transport = repeated ? MXLO_SHARD_LOOPBACK : MXLO_SHARD_RCCL;
  if (transport == MXLO_SHARD_RCCL && repeated) {
    set_serr("mxlo_shard_ctx_create: RCCL refuses several ranks on one device — use the loopback or the peer transport for repeated device ids");
    delete s;
    return MXLO_EINVAL;
  }
  if (transport == MXLO_SHARD_LOOPBACK && !repeated && ndev > 1) {
    set_serr("mxlo_shard_ctx_create: the loopback transport sums on ONE device — list one device id %d times", ndev);
    delete s;
    return MXLO_EINVAL;
  }
  s->transport = transport;
  s->loopback = transport == MXLO_SHARD_LOOPBACK;
  s->same_device = repeated || ndev == 1;
  s->uses_bar = s->loopback || (transport == MXLO_SHARD_PEER && s->same_device);
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
  if (st == MXLO_OK && transport == MXLO_SHARD_RCCL) {
    s->comms.assign(ndev, nullptr);
    const char *inject = getenv("MXLO_SHARD_FAULT");      // TEST HOOK: "initall" = ncclCommInitAll reports an error
    ncclResult_t r = (inject && !strcmp(inject, "initall")) ? ncclSystemError : ncclCommInitAll(s->comms.data(), ndev, s->dev.data());
    if (r != ncclSuccess) {
      set_serr("ncclCommInitAll over %d device(s): %s", ndev, ncclGetErrorString(r));
      for (auto &c : s->comms) c = nullptr;
      st = MXLO_EREDUCE;
    } else {
      for (int i = 0; i < ndev && st == MXLO_OK; ++i) {   // what the communicators themselves say: N ranks, rank i on device i
        int cnt = -1, ur = -1, cd = -1;
        if (ncclCommCount(s->comms[i], &cnt) != ncclSuccess || ncclCommUserRank(s->comms[i], &ur) != ncclSuccess ||
            ncclCommCuDevice(s->comms[i], &cd) != ncclSuccess || cnt != ndev || ur != i || cd != s->dev[i]) {
          set_serr("ncclCommInitAll: communicator %d reports %d ranks / rank %d / device %d, expected %d / %d / %d", i, cnt, ur, cd,
                   ndev, i, s->dev[i]);
          s->poisoned = true;                            // destroy aborts the communicators
          st = MXLO_EREDUCE;
        }
      }
    }
  }
  if (st == MXLO_OK && s->uses_bar) s->bar.n = ndev;
  if (st == MXLO_OK && s->loopback) {
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
  if (st == MXLO_OK && transport == MXLO_SHARD_PEER) st = peer_setup(s);
  s->hook_fn = s->loopback ? loop_hook : (transport == MXLO_SHARD_PEER ? peer_hook : rccl_hook);
  for (int i = 0; i < ndev && st == MXLO_OK; ++i) st = mxlo_ctx_set_allreduce(s->ctx[i], s->hook_fn, &s->users[i]);
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

API int32_t mxlo_shard_ctx_create(int32_t ndev, const int32_t *dev_ids, mxlo_shard_ctx **out) {
  return mxlo_shard_ctx_create_ex(ndev, dev_ids, MXLO_SHARD_AUTO, out);
}

API int32_t mxlo_shard_ctx_transport(mxlo_shard_ctx *s) { return s ? s->transport : 0; }

// What shard i runs on: device ordinal, its PCI bus id, and how many ranks the transport itself reports (ncclCommCount
// for RCCL; the shard count otherwise).
API int32_t mxlo_shard_ctx_info(mxlo_shard_ctx *s, int32_t i, int32_t *device, int32_t *ranks_seen, char *pci_bus_id, int32_t pci_len) {
  if (!s || i < 0 || i >= s->ndev) {
    set_serr("mxlo_shard_ctx_info: bad shard index");
    return MXLO_EINVAL;
  }
  if (device) *device = s->dev[i];
  if (ranks_seen) {
    int cnt = s->ndev;
    if (s->transport == MXLO_SHARD_RCCL && (ncclCommCount(s->comms[i], &cnt) != ncclSuccess)) cnt = -1;
    *ranks_seen = cnt;
  }
  if (pci_bus_id && pci_len > 0) {
    pci_bus_id[0] = 0;
    if (hipDeviceGetPCIBusId(pci_bus_id, pci_len, s->dev[i]) != hipSuccess) {
      (void)hipGetLastError();
      snprintf(pci_bus_id, (size_t)pci_len, "?");
    }
  }
  return MXLO_OK;
}

// TEST HOOKS of the peer transport: "peer_drop" = shard index that never posts (-1: none), "peer_timeout_ms".
API int32_t mxlo_shard_ctx_debug(mxlo_shard_ctx *s, const char *key, int64_t value) {
  if (!s || !key) return MXLO_EINVAL;
  if (!strcmp(key, "peer_drop")) {
    for (auto &c : s->pc) c.drop = (int)value;
  } else if (!strcmp(key, "peer_timeout_ms")) {
    for (auto &c : s->pc) c.timeout_ms = (int)value;
  } else {
    set_serr("mxlo_shard_ctx_debug: unknown key '%s'", key);
    return MXLO_EINVAL;
  }
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
    if (e != MXLO_OK && st == MXLO_OK) {
      st = e;
      set_serr("shard %d (device %d): %s", i, s->dev[i], mxlo_last_error());
    }
  }
  for (int i = 0; i < (int)s->pc.size(); ++i)     // a peer gather that gave up has stored NaN: say so
    if (mxlo_peer::comm_fault(&s->pc[i]) != 0) {
      s->poisoned = true;
      if (st == MXLO_OK) {
        st = MXLO_EREDUCE;
        set_serr("shard %d (device %d): %s", i, s->dev[i], mxlo_peer::last_error());
      }
    }
  return st;
}

// The transport's preflight (include/mxlo_rccl.h: mxlo_rccl_preflight_hook) on every shard of the ctx, through the hook
// the `_sharded` entry points use. latency_us[k] = the slowest shard's latency for payload k.
API int32_t mxlo_shard_ctx_preflight(mxlo_shard_ctx *s, int32_t reps, int32_t timeout_ms, double latency_us[3]) {
  if (!s || !latency_us) {
    set_serr("mxlo_shard_ctx_preflight: NULL argument");
    return MXLO_EINVAL;
  }
  std::vector<double> lat((size_t)s->ndev * 3, 0.0);
  std::vector<std::string> errs(s->ndev);
  const int32_t st = run_all(s, [&](int i) -> int32_t {
    const int32_t rc = mxlo_rccl_preflight_hook(s->hook_fn, &s->users[i], s->transport == MXLO_SHARD_RCCL ? (void *)s->comms[i] : nullptr, i,
                                                s->ndev, (void *)s->streams[i], reps, timeout_ms, &lat[(size_t)i * 3]);
    if (rc != 0) {
      errs[i] = mxlo_rccl_last_error();
      return MXLO_EREDUCE;
    }
    return MXLO_OK;
  });
  if (st != MXLO_OK) {
    for (int i = 0; i < s->ndev; ++i)
      if (!errs[i].empty()) {
        set_serr("shard %d (device %d): %s", i, s->dev[i], errs[i].c_str());
        break;
      }
    return st;
  }
  for (int k = 0; k < 3; ++k) {
    latency_us[k] = 0.0;
    for (int i = 0; i < s->ndev; ++i) latency_us[k] = lat[(size_t)i * 3 + k] > latency_us[k] ? lat[(size_t)i * 3 + k] : latency_us[k];
  }
  return MXLO_OK;
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
