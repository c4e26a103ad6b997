// peer.hip — libmxlo_rccl.so: the peer-mapped ONE-SHOT scalar exchange (SURVEY.md §5, §8e last paragraph), a second
// transport of the all-reduce hook beside RCCL.
//
// Every collective of the hot path sums 1 .. 864 doubles (§8e): latency-bound. An RCCL all-reduce of that size costs a
// proxy round trip and a ring of kernel-side handshakes; here each rank owns a MAILBOX that every other rank can store
// into, and one collective is ONE small kernel per rank:
//   post   : the rank stores its `count` doubles into ITS slot of every rank's mailbox (system-scope stores; over xGMI for
//            peer-mapped device memory, over PCIe for host memory), fences, then stores the collective's sequence number
//            behind them (release);
//   gather : it polls the `world` sequence words of its OWN mailbox (bounded wait), then adds the `world` payloads in
//            FIXED RANK ORDER — every rank adds the same values in the same order, so all ranks hold identical bits (what
//            the replicated control flow of push! / the ys[k] != 0 skips needs) — and writes the sums over its input.
// Two slot sets alternate by the parity of the sequence number: a rank can post collective k + 2 only after it has
// gathered k + 1, which needs every rank's post k + 1, which every rank issues after its own gather k (stream order) — so
// set k & 1 is never overwritten before everybody has read it. No re-arming, no RCCL call, no host round trip.
//
// Mailbox memory (whoever creates the comm decides):
//   * single process, several devices (shard.hip, MXLO_SHARD_PEER): fine-grained device memory + hipDeviceEnablePeerAccess
//     (the one-shot gather over the fully connected xGMI links of §5), or pinned host memory when a pair of devices has
//     no peer access;
//   * one process per GPU (mxlo_peer_comm_create_shm): a POSIX shared-memory segment registered with every process's HIP
//     runtime (hipHostRegister, mapped): coherent by construction, no hipIpc / dmabuf dependency.
// A rank whose peers never post (a dead process, a hung device) does not hang the GPU: after `timeout_ms` of polling the
// kernel stores NaN, raises the comm's fault word (pinned host memory) and ENDS; the next hook call / sync returns an error.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/mxlo.h"
#include "../../include/mxlo_rccl.h"
#include "peer.h"

#define API extern "C" __attribute__((visibility("default")))

namespace mxlo_peer {

static thread_local char g_perr[512] = "";
const char *last_error() { return g_perr; }
static int32_t pfail(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
static int32_t pfail(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_perr, sizeof(g_perr), fmt, ap);
  va_end(ap);
  return 1;
}

struct KArgs {
  unsigned long long *mb[kMaxPeers];   // rank r's mailbox, as THIS device addresses it
  int rank, world, drop;
  unsigned long long ticks;            // wall_clock64() ticks of the bounded wait
  unsigned *fault;                     // device address of the comm's pinned fault word
  unsigned long long *seq_dev;         // this rank's sequence counter (device memory, touched by this kernel only)
};

__device__ __forceinline__ void st_sys(unsigned long long *p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned long long ld_sys(const unsigned long long *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// phase: 1 = post, 2 = gather, 3 = both (one launch per collective)
__global__ void __launch_bounds__(256)
peer_exchange_kernel(KArgs A, double *__restrict__ buf, int count, int phase) {
  const int tid = threadIdx.x, me = A.rank, world = A.world;
  // The sequence number of this collective: the posting half advances the rank's device counter, the gathering half of a
  // two-launch collective reads what the posting launch left (same stream: ordered). Every rank runs one (pair of)
  // launch(es) per collective, eager or replayed from a graph, so the counters advance in lockstep.
  __shared__ unsigned long long seq_sh;
  if (tid == 0) {
    unsigned long long s = *A.seq_dev;
    if (phase & 1) *A.seq_dev = ++s;
    seq_sh = s;
  }
  __syncthreads();
  const unsigned long long seq = seq_sh;
  const size_t set_off = (size_t)(seq & 1ull) * (size_t)world * kSlotWords;
  if (phase & 1) {
    if (me != A.drop) {
      for (int idx = tid; idx < world * count; idx += 256) {
        const int d = idx / count, t = idx - d * count;
        st_sys(A.mb[d] + set_off + (size_t)me * kSlotWords + kHeaderWords + t, (unsigned long long)__double_as_longlong(buf[t]));
      }
      __threadfence_system();            // every payload store of this lane is visible system-wide ...
      __syncthreads();                   // ... for ALL lanes, before any sequence word is published: the publishing store
      if (tid < world)                   // itself can then be relaxed (one fence per lane, not two)
        __hip_atomic_store(A.mb[tid] + set_off + (size_t)me * kSlotWords, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  if (phase & 2) {
    __shared__ int bad;
    if (tid == 0) bad = 0;
    __syncthreads();
    if (tid < world) {
      const unsigned long long *w = A.mb[me] + set_off + (size_t)tid * kSlotWords;
      unsigned long long t0 = 0;
      unsigned it = 0;
      while (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {   // relaxed polls, ONE acquire fence below
        __builtin_amdgcn_s_sleep(2);
        if ((++it & 63u) == 0) {
          const unsigned long long now = (unsigned long long)wall_clock64();
          if (t0 == 0) t0 = now;
          else if (now - t0 > A.ticks) {
            bad = 1;
            __hip_atomic_store(A.fault, 0x100u + (unsigned)tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // which rank was missing
            break;
          }
        }
      }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);   // (system scope) the payloads behind the sequence words seen above
    __syncthreads();
    const bool failed = bad != 0;
    for (int t = tid; t < count; t += 256) {
      const unsigned long long *p = A.mb[me] + set_off + kHeaderWords + t;
      double s = __longlong_as_double((long long)ld_sys(p));
      for (int r = 1; r < world; ++r) s += __longlong_as_double((long long)ld_sys(p + (size_t)r * kSlotWords));   // rank order
      buf[t] = failed ? __longlong_as_double(0x7FF8000000000000ll) : s;
    }
  }
}

int32_t comm_init_common(Comm *c, int rank, int world, int timeout_ms) {
  if (world < 1 || world > kMaxPeers || rank < 0 || rank >= world) return pfail("peer comm: rank %d not in [0, %d), world <= %d", rank, world, kMaxPeers);
  c->rank = rank;
  c->world = world;
  c->timeout_ms = timeout_ms > 0 ? timeout_ms : 30000;
  int dev = 0, khz = 0;
  if (hipGetDevice(&dev) != hipSuccess) return pfail("peer comm: hipGetDevice failed");
  c->device = dev;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000;
  c->wall_khz = khz;
  void *hp = nullptr, *dp = nullptr;
  if (hipHostMalloc(&hp, 64, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess) {
    if (hp) (void)hipHostFree(hp);
    return pfail("peer comm: no mapped pinned memory for the fault word");
  }
  memset(hp, 0, 64);
  c->fault_host = (unsigned *)hp;
  c->fault_dev = (unsigned *)dp;
  if (hipMalloc((void **)&c->seq_dev, 64) != hipSuccess || hipMemset(c->seq_dev, 0, 64) != hipSuccess) {
    (void)hipGetLastError();
    if (c->seq_dev) (void)hipFree(c->seq_dev);
    c->seq_dev = nullptr;
    (void)hipHostFree(hp);
    c->fault_host = c->fault_dev = nullptr;
    return pfail("peer comm: no device memory for the sequence counter");
  }
  return 0;
}

void comm_release_common(Comm *c) {
  if (c->fault_host) (void)hipHostFree(c->fault_host);
  c->fault_host = c->fault_dev = nullptr;
  if (c->seq_dev) (void)hipFree(c->seq_dev);
  c->seq_dev = nullptr;
}

int32_t comm_fault(Comm *c) {
  if (!c->fault_host) return 0;
  const unsigned code = __atomic_load_n(c->fault_host, __ATOMIC_RELAXED);
  if (code == 0) return 0;
  c->dead = true;
  return pfail("peer exchange: rank %d waited %d ms for rank %u's posting (about this rank's %llu-th launched collective) and gave up (that rank is gone, "
               "stuck, or not part of this communicator); the results of that collective are NaN and this communicator is unusable",
               c->rank, c->timeout_ms, code - 0x100u, (unsigned long long)c->seq);
}

// phase 3: the whole collective in one launch; 1 / 2: the two halves (same-device shards put a host barrier between them)
int32_t comm_launch(Comm *c, double *buf, int64_t count, hipStream_t st, int phase) {
  if (count <= 0) return 0;
  if (count > kCap) return pfail("peer exchange: %lld doubles exceed the mailbox slot of %d", (long long)count, kCap);
  KArgs A;
  for (int r = 0; r < kMaxPeers; ++r) A.mb[r] = r < c->world ? c->mb[r] : nullptr;
  A.rank = c->rank;
  A.world = c->world;
  A.drop = c->drop;
  A.ticks = (unsigned long long)c->timeout_ms * (unsigned long long)c->wall_khz;
  A.fault = c->fault_dev;
  A.seq_dev = c->seq_dev;
  hipLaunchKernelGGL(peer_exchange_kernel, dim3(1), dim3(256), 0, st, A, buf, (int)count, phase);
  if (hipGetLastError() != hipSuccess) return pfail("peer exchange: kernel launch failed");
  return 0;
}

int32_t comm_allreduce(Comm *c, double *buf, int64_t count, hipStream_t st) {
  if (c->dead) return pfail("peer exchange: this communicator is unusable after a timed-out collective");
  if (comm_fault(c) != 0) return 1;
  if (count <= 0) return 0;
  ++c->seq;
  return comm_launch(c, buf, count, st, 3);
}

}  // namespace mxlo_peer

using namespace mxlo_peer;

// ---- one process per GPU: mailboxes in a POSIX shared-memory segment ---------------------------------------------------
struct mxlo_peer_comm {
  Comm c;
  std::string name;
  void *map = nullptr;
  size_t map_bytes = 0;
  bool registered = false, owner = false;
};

API const char *mxlo_peer_last_error(void) { return mxlo_peer::last_error(); }

// Collective over the host runtime: rank 0 calls with create = 1 FIRST (the segment is created exclusively and zero-filled:
// sequence number 0 never matches a collective), the other ranks after rank 0 has returned (a host-runtime barrier).
// `name` is a POSIX shm name ("/mxlo-<token>"), unique per job. The CURRENT HIP device is the rank's device.
API int32_t mxlo_peer_comm_create_shm(const char *name, int32_t rank, int32_t world, int32_t create, int32_t timeout_ms,
                                      mxlo_peer_comm **out) {
  if (!name || name[0] != '/' || !out) return pfail("mxlo_peer_comm_create_shm: bad argument (the name must start with '/')");
  *out = nullptr;
  mxlo_peer_comm *p = new mxlo_peer_comm();
  if (comm_init_common(&p->c, rank, world, timeout_ms) != 0) {
    delete p;
    return 1;
  }
  p->name = name;
  p->owner = false;                    // set once the EXCLUSIVE create has succeeded: a name that already exists belongs to
                                       // somebody else's job and must not be unlinked by this one's error path
  const size_t bytes = (size_t)world * mailbox_words(world) * sizeof(unsigned long long);
  const int fd = create ? shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600) : shm_open(name, O_RDWR, 0600);
  auto bail = [&](const char *what) {
    pfail("mxlo_peer_comm_create_shm(%s, rank %d): %s failed: %s", name, rank, what, strerror(errno));
    if (fd >= 0) close(fd);
    if (p->map) munmap(p->map, p->map_bytes);
    if (p->owner) shm_unlink(name);
    comm_release_common(&p->c);
    delete p;
    return 1;
  };
  if (fd < 0) return bail(create ? "shm_open(O_CREAT|O_EXCL)" : "shm_open");
  p->owner = create != 0;
  if (create && ftruncate(fd, (off_t)bytes) != 0) return bail("ftruncate");
  if (!create) {
    struct stat sb;
    if (fstat(fd, &sb) != 0 || (size_t)sb.st_size < bytes) {
      errno = EINVAL;
      return bail("size check (rank 0 has not created the segment for this world size yet)");
    }
  }
  void *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  if (m == MAP_FAILED) return bail("mmap");
  p->map = m;
  p->map_bytes = bytes;
  close(fd);
  if (hipHostRegister(m, bytes, hipHostRegisterMapped | hipHostRegisterPortable) != hipSuccess) {
    (void)hipGetLastError();
    pfail("mxlo_peer_comm_create_shm(%s, rank %d): hipHostRegister of the %zu-byte segment failed", name, rank, bytes);
    munmap(m, bytes);
    if (p->owner) shm_unlink(name);
    comm_release_common(&p->c);
    delete p;
    return 1;
  }
  p->registered = true;
  void *dptr = nullptr;
  if (hipHostGetDevicePointer(&dptr, m, 0) != hipSuccess) {
    (void)hipGetLastError();
    pfail("mxlo_peer_comm_create_shm: hipHostGetDevicePointer failed");
    (void)hipHostUnregister(m);
    munmap(m, bytes);
    if (p->owner) shm_unlink(name);
    comm_release_common(&p->c);
    delete p;
    return 1;
  }
  for (int r = 0; r < world; ++r) p->c.mb[r] = (unsigned long long *)dptr + (size_t)r * mailbox_words(world);
  *out = p;
  return 0;
}

API int32_t mxlo_peer_comm_destroy(mxlo_peer_comm *p) {
  if (!p) return 0;
  if (p->registered) (void)hipHostUnregister(p->map);
  if (p->map) munmap(p->map, p->map_bytes);
  if (p->owner) shm_unlink(p->name.c_str());
  comm_release_common(&p->c);
  delete p;
  return 0;
}

// an `mxlo_allreduce_fn`: user = the mxlo_peer_comm; ONE kernel on `stream` per collective
API int32_t mxlo_peer_allreduce_hook(void *user, void *dev_buf, int64_t count, void *stream) {
  if (!user || !dev_buf || count < 0) return 1;
  return comm_allreduce(&((mxlo_peer_comm *)user)->c, (double *)dev_buf, count, (hipStream_t)stream);
}

// 0 = healthy; non-zero (message in mxlo_peer_last_error) after a collective of this rank timed out. Reads pinned memory:
// call it after the stream has been synchronised to learn whether the LAST collective went through.
API int32_t mxlo_peer_comm_check(mxlo_peer_comm *p) { return p ? comm_fault(&p->c) : 1; }

// test hooks: "drop" = this rank never posts (value 1) ; "timeout_ms"
API int32_t mxlo_peer_comm_debug(mxlo_peer_comm *p, const char *key, int64_t value) {
  if (!p || !key) return 1;
  if (!strcmp(key, "drop")) p->c.drop = value ? p->c.rank : -1;
  else if (!strcmp(key, "timeout_ms")) p->c.timeout_ms = (int)value;
  else return pfail("mxlo_peer_comm_debug: unknown key '%s'", key);
  return 0;
}
