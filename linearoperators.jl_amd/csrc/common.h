// common.h — internal definitions shared by the libmxlo.so translation units.
// gfx950 (MI355X, CDNA4) only: wave = 64 lanes, 256 CUs in 8 XCDs.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../include/mxlo.h"

#define MXLO_API extern "C" __attribute__((visibility("default")))

namespace mxlo {

constexpr int kBlock = 256;          // 4 waves of 64 lanes: one wave per SIMD of a CU
constexpr int kWave = 64;
constexpr int kMaxRedCols = 128;     // columns a single reduction call may produce
constexpr int kQnfMaxCols = 40, kQnfMaxGrid = 256;   // single-launch quasi-Newton apply: panel columns x workgroups of its exchange
constexpr int kQnfSlots = kQnfMaxCols * kQnfMaxGrid;
constexpr int kFusedSlots = 512;     // workgroups of a single-launch (grid-exchange) kernel: all co-resident (round 6: up to two per CU)
constexpr unsigned long long kSlotEmpty = 0x7FF8DEADBEEF0001ull;   // a NaN payload no arithmetic produces (partials are canonicalised)
constexpr int kMaxRedBlocks = 4096;  // partial slots per column in the workspace
constexpr int kScalarSlots = 8192;   // doubles in the device scalar buffer

void set_error(const char *fmt, ...);

// ---- runtime-call accounting (the f2 contract: a warmed mul!/diag!/solve allocates nothing, copies nothing, never
// synchronises) -----------------------------------------------------------------------------------------------
// Every HIP runtime entry point the library uses that allocates, copies or blocks goes through these macros (a
// function-like macro is not re-expanded inside its own replacement, so the inner name is the real HIP function);
// kernel launches are counted in MXLO_LAUNCH_CHECK. mxlo_debug_counters() exports the totals for the contract test.
struct ApiCounters {
  std::atomic<int64_t> n_malloc{0}, n_free{0}, n_h2d{0}, n_d2h{0}, n_d2d{0}, n_d2h_bytes{0}, n_stream_sync{0},
      n_device_sync{0}, n_event_sync{0}, n_memset_async{0}, n_launch{0}, n_blocking_copy{0};
};
inline ApiCounters &api_counters() {   // one instance per linked image (the library's is what mxlo_debug_counters reports)
  static ApiCounters c;
  return c;
}
inline void count_copy(size_t bytes, hipMemcpyKind kind, bool blocking) {
  ApiCounters &c = api_counters();
  if (kind == hipMemcpyDeviceToHost) { ++c.n_d2h; c.n_d2h_bytes += (int64_t)bytes; }
  else if (kind == hipMemcpyHostToDevice) ++c.n_h2d;
  else ++c.n_d2d;
  if (blocking) ++c.n_blocking_copy;
}
inline hipError_t counted_memcpy(void *d, const void *s, size_t b, hipMemcpyKind k) {
  count_copy(b, k, true);
  return hipMemcpy(d, s, b, k);
}
inline hipError_t counted_memcpy_async(void *d, const void *s, size_t b, hipMemcpyKind k, hipStream_t st) {
  count_copy(b, k, false);
  return hipMemcpyAsync(d, s, b, k, st);
}
#define hipMalloc(...) (++mxlo::api_counters().n_malloc, hipMalloc(__VA_ARGS__))
#define hipFree(...) (++mxlo::api_counters().n_free, hipFree(__VA_ARGS__))
#define hipHostMalloc(...) (++mxlo::api_counters().n_malloc, hipHostMalloc(__VA_ARGS__))
#define hipHostFree(...) (++mxlo::api_counters().n_free, hipHostFree(__VA_ARGS__))
#define hipMemcpy(...) mxlo::counted_memcpy(__VA_ARGS__)
#define hipMemcpyAsync(...) mxlo::counted_memcpy_async(__VA_ARGS__)
#define hipStreamSynchronize(...) (++mxlo::api_counters().n_stream_sync, hipStreamSynchronize(__VA_ARGS__))
#define hipDeviceSynchronize(...) (++mxlo::api_counters().n_device_sync, hipDeviceSynchronize(__VA_ARGS__))
#define hipEventSynchronize(...) (++mxlo::api_counters().n_event_sync, hipEventSynchronize(__VA_ARGS__))
#define hipMemsetAsync(...) (++mxlo::api_counters().n_memset_async, hipMemsetAsync(__VA_ARGS__))

#define MXLO_HIP(call)                                                                           \
  do {                                                                                           \
    hipError_t e__ = (call);                                                                     \
    if (e__ != hipSuccess) {                                                                     \
      mxlo::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
      return MXLO_EHIP;                                                                          \
    }                                                                                            \
  } while (0)

#define MXLO_REQUIRE(cond, code, ...)                                                            \
  do {                                                                                           \
    if (!(cond)) {                                                                               \
      mxlo::set_error(__VA_ARGS__);                                                              \
      return (code);                                                                             \
    }                                                                                            \
  } while (0)

#define MXLO_TRY(expr)                                                                           \
  do {                                                                                           \
    int32_t s__ = (expr);                                                                        \
    if (s__ != MXLO_OK) return s__;                                                              \
  } while (0)

// MXLO_TRACE=1 in the environment: synchronise after every launch and log its source line (a GPU memory fault
// aborts the process without an error code; the last line logged is the launch that faulted).
inline bool mxlo_trace_on() {
  static const bool on = getenv("MXLO_TRACE") != nullptr;
  return on;
}
#define MXLO_LAUNCH_CHECK()                                                                      \
  do {                                                                                           \
    ++mxlo::api_counters().n_launch;                                                             \
    if (mxlo_trace_on()) {                                                                       \
      fprintf(stderr, "[mxlo] launched %s:%d ... ", __FILE__, __LINE__);                         \
      fflush(stderr);                                                                            \
      (void)hipDeviceSynchronize();                                                              \
      fprintf(stderr, "done\n");                                                                 \
    }                                                                                            \
    hipError_t e__ = hipGetLastError();                                                          \
    if (e__ != hipSuccess) {                                                                     \
      mxlo::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__), __FILE__,      \
                      __LINE__);                                                                 \
      return MXLO_EHIP;                                                                          \
    }                                                                                            \
  } while (0)

struct Tune {
  int blocks_per_cu = 0;   // streaming kernels: 0 = one chunk per workgroup; k = persistent grid CUs*k
  int extend_tiles_per_block = 0;     // sorted extension: consecutive output tiles per workgroup (0 = auto)
  int64_t nt_min_bytes = 256ll << 20;  // streamed footprint from which nontemporal accesses are used: the size of the
                                       // Infinity Cache — below it the second pass of a two-pass apply finds the first
                                       // pass's lines there (profiles/r03_sweep_nt_mid.txt: -5 ... -8 % at n = 2^21 .. 2^23)
  int red_blocks_per_cu = 4;  // reduction kernels
  int graph_direct_max = 16; // captured chains of at most this many kernel/memset nodes replay as direct launches
  int house_fused_per_cu = 2;   // single-launch Householder: workgroups per CU it may use (2, round 6: vectors up to 2^22 doubles; 1: up to 2^21)
  int house_fused = 1;     // single-launch Householder (dot, grid exchange, update) while the vectors fit one wave of workgroups
  int cherm_two_pass = 0;  // complex opHermitian: 1 = the two-pass (rows, then columns) form instead of the strip kernel
  int house_reverse = 1;   // Householder phase B walks the vectors back-to-front (MALL tail reuse)
  int64_t house_inline_n = 1ll << 23;   // two-pass Householder up to this n: the update pass sums the dots pass's partials
                                        // itself (no finalize launch); above, one finalize launch is cheaper than every
                                        // update workgroup re-reading the partials
  int lbfgs_inv_mode = MXLO_INV_TWOPASS;
  int dots_max_nc = 20;    // columns per panel_dots launch (<= 20)
  int gemm_tile = 0;       // kron GEMM tile edge: 0 = auto (largest of 128/64/32 that still gives every CU a
                           // workgroup), 32 / 64 / 128 force one, -1 = generic fallback kernel only
  int fuse_finalize = 1;   // reductions of <= 4 columns: last-arriving workgroup finalizes in the dots kernel
  int combine_blocks_per_cu = 0;   // panel_combine: 0 = one vector per thread (best measured), k = persistent grid
  int qn_fused_max_grid = 256;  // single-launch quasi-Newton apply: most workgroups it may use (<= 256; 64 = the round-3 limit)
  int qn_fused_batch12 = 1;   // single-launch quasi-Newton apply with 9 .. 12 columns on short vectors: all columns in one batch
  int qn_fused_small = 1;  // quasi-Newton applies with <= 64 workgroups of dots: dots + finalize + coefficients in one launch
  int qn_persist = 1;      // quasi-Newton applies at cache-resident sizes: ONE persistent launch (one workgroup per CU, grid
                           // exchange between the dots and the combine phase; qn.hip: qn_apply_persist_kernel)
  int64_t qn_persist_min_n = 1ll << 19;          // ... from this vector length on (below: the single-launch slice form)
  int64_t qn_persist_max_bytes = 448ll << 20;    // ... while the panel (columns x n x element size) is at most this many bytes
                                                 // (profiles/r05_tune_persist.txt: ahead of four launches up to ~340 MB panels)
  int qn_persist_reverse = 1;   // ... its combine phase walks the workgroup's chunks back to front
  int qn_persist_prefetch = 0;  // ... x and the first column batch of its first combine chunk are requested before the exchange.
                                // Measured (profiles/r05_bench_mid_apply.txt, last block): 1.5 us SLOWER at n = 2^19 .. 2^20 (the polls of
                                // the exchange return in order behind the 11 prefetch loads), neutral above — off
  int qn_persist_lds = 1;       // ... x and the first columns of the combine order parked in the CU's LDS by the dots phase (round 6)
  int qn_persist_lds_pad = 0;   // ... bytes of (unused) dynamic LDS requested per workgroup: > 80 KiB forces one workgroup per CU
  int64_t qn_persist_min_bytes = 32ll << 20;     // ... and at least this many (an L-SR1 m = 5 apply at n = 2^19 — 21 MB — is
                                                 // faster in the single-launch slice form: 10.7 vs 12.1 us)
  int herm_order = 1;      // opHermitian interior strips: 0 = row group by row group, 1 (default, round 6) = column block by column block (dense.hip)
  int herm_nt = -1;        // opHermitian strip loads: -1 = nontemporal except for triangles of [herm_dp_min_bytes, herm_nt_min_bytes)
                           // (about the size of the Infinity Cache; round 6, dense.hip: herm_nt_policy), 0 / 1 force
  int64_t herm_dp_min_bytes = 96ll << 20;
  int64_t herm_nt_min_bytes = 384ll << 20;
  int herm_poll_sleep = 4;    // single-launch opHermitian: the finishers' first wait looks every 64 x this many clocks (round 5: 32;
                              // sweep 1 ... 64 in profiles/r06_herm_policy.txt: 2 ... 8 are 0 ... 5 % ahead of 32 at n = 2048 / 3072 / f32 6144, equal elsewhere)
  int herm_strip = 0;      // opHermitian: tiles per strip, 0 = by size (8 once there are two 8-tile strips per CU, else 2, else 1); 1 / 2 / 8 force (sweeps)
  int herm_lds_pad = 0;    // opHermitian pass launch: bytes of unused dynamic LDS per workgroup (occupancy experiment, <= 48 KiB)
  int herm_single = 1;     // opHermitian (full row groups, aligned A, n <= herm_single_max_n): strips and finishers in ONE launch
  int64_t herm_single_max_n = 0;      // 0 (default): by size — triangles of at most herm_single_max_bytes; > 0: n <= this value
  int64_t herm_single_max_bytes = 112ll << 20;   // measured (profiles/r06_herm_policy.txt, round 6): f64 gains up to n = 5120
                                      // (4096: 15.2 -> 14.8 us, 5120: 20.4 -> 19.8), equal at 3072, LOSES at 6144 (28.3 -> 28.8); f32 gains up
                                      // to 6144 (5120: 15.5 -> 14.1, 6144: 18.5 -> 17.6), equal at 8192. Round 5 measured no gain above
                                      // 2048: that was with the row-group order and the three-round finish.
  int kron_fuse = 1;       // kron: both GEMMs in ONE launch when every tile has its own CU, the dependency kept XCD-local
                           // (gemm_glds.h: kron_fused_kernel; 2: timing experiment without the wait — wrong results)
  int gemv_n_rows = 1;     // dense M*v: row bands, the column sum stays inside a workgroup — one launch, no partials (dense.hip)
  int gemvb_n_rows = 1;    // block apply M*V of a dense operator: row bands, V staged in LDS per workgroup — one launch, no partials (dense.hip)
  int gemvb_t_lds = 1;     // transposed block apply of a dense operator (k >= 4): U staged in LDS per workgroup (dense.hip)
  int combine_reverse = 0; // four-launch applies: the combine pass walks the vectors back to front. Measured (round 5,
                           // profiles/r05_bench_mid_apply.txt): -3.6 … +2.8 %, no gain on average — the grid-stride dots pass
                           // leaves no usable tail behind; the persistent launch (whose workgroups own contiguous runs) does
  int push_wide = 1;       // one-pass push!: 20 columns per pass while >= 20 remain (0: always <= 10)
  int push_posted = 1;     // push!'s decision scalars: posted into mapped pinned host memory by a one-wave kernel and polled
                           // by the host (1) or copied with hipMemcpyAsync + a stream synchronisation (0)
  int push_fused = 1;      // push!(op, s, y): one-pass schedule (new pair held per lane, in-pass slot stores); 0 = copies + dual-x dots
  int sp_xcds = 1;         // sparse apply: XCDs (L2 domains) the chunk order is banded over (8: XCD k walks the k-th contiguous
                           // eighth of the chunk table); 1 = plain order, the default: banding measured -8 % … +8 % by pattern
  int fused_timeout_ms = 2000;   // single-launch (grid-exchange) kernels: how long a workgroup polls for its peers' partials
                                 // before it gives up, raises the ctx fault flag and stores NaN (a launch that is not fully
                                 // co-resident — GPU shared with other processes, CU masking — ends instead of hanging)
  int fused_debug_drop = -1;     // TEST HOOK: this workgroup of a single-launch kernel never publishes its partial
};

}  // namespace mxlo

struct mxlo_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  int num_cu = 256;
  double *partials = nullptr;  // [kMaxRedCols][kMaxRedBlocks] per-block partial sums
  double *scalars = nullptr;   // finalized reduction results / device-resident coefficients
  bool lds_attr_set = false;   // dynamic-LDS limits of the shifted-solve kernels raised on THIS device
  bool inv_lds_attr_set = false;   // ... and inv_coef_kernel at na = 64
  unsigned *ticket = nullptr;  // arrival counter of the fused (last-workgroup) finalize; zero between launches
  unsigned long long *xslots = nullptr;  // [2][kFusedSlots] partial-exchange slots + epoch word of the single-launch Householder
  unsigned long long *qslots = nullptr;  // [2][40 x 64] exchange slots + epoch word of the single-launch quasi-Newton apply (qn.hip)
  unsigned *fault_host = nullptr;        // pinned, device-mapped word: a single-launch kernel that timed out on its exchange
  unsigned *fault_dev = nullptr;         // stores its code here (system scope); the host reads it without synchronising
  int wall_clock_khz = 100000;           // rate of wall_clock64() on this device (hipDeviceAttributeWallClockRate)
  mxlo_allreduce_fn allreduce = nullptr;
  void *allreduce_user = nullptr;
  double *herm_slots = nullptr;   // self-validating partial slots of the single-launch opHermitian (all empty between applies)
  size_t herm_slots_bytes = 0;
  int64_t herm_slots_layout = 0;  // (n, strip shape) the slots were last used with
  bool herm_slots_dirty = false;  // a timed-out apply left slots filled: re-arm before the next use
  unsigned *kron_cnt = nullptr;   // counters of the XCD-local kron fusion (gemm_glds.h: kron_fused_kernel), zero between applies
  int kron_cnt_n = 0;
  int xcd_map = 0;                // 0: not probed; 1: workgroup id % 8 == XCC_ID on this device (probed once); -1: it is not
  void *scratch = nullptr;     // grow-on-demand workspace (opHermitian tile partials); owned by the ctx
  size_t scratch_bytes = 0;
  hipStream_t own_stream = nullptr;  // created by mxlo_ctx_create_stream, destroyed with the ctx
  hipEvent_t switch_event = nullptr; // orders the ctx workspaces across mxlo_ctx_set_stream
  bool capturing = false;      // between mxlo_graph_begin and mxlo_graph_end
  std::vector<std::pair<mxlo_qn *, int64_t>> captured_qn;   // quasi-Newton handles (and their generation) applied inside the open capture
  int64_t scratch_generation = 0;   // bumped whenever `scratch` is reallocated (captured opHermitian applies go stale)
  bool scratch_used_in_capture = false;
  mxlo::Tune tune;
};

namespace mxlo {

// Every entry point that allocates or launches runs on the device its ctx was created for, whatever the calling
// thread's current device is (one process driving several GPUs), and leaves the caller's current device untouched.
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard &) = delete;
  DeviceGuard &operator=(const DeviceGuard &) = delete;
};
#define MXLO_DEVICE_GUARD(ctxexpr) mxlo::DeviceGuard dev_guard__((ctxexpr)->device)

// ---- single-launch (grid-exchange) kernels: bounded wait + fault flag -----------------------------------------------
// The workgroups of householder_fused_kernel / qn_apply_fused_kernel wait for each other's partials, so the whole grid
// must be resident at once. The launch sites check that against the occupancy of the kernel (coresident()); what no
// host-side check can see — other processes occupying the CUs, CU masking, a launch killed half-way that left the
// slots inconsistent — is caught by the wait itself: a lane that has polled one slot for longer than
// `fused_timeout_ms` raises the ctx fault word (pinned host memory, system-scope store) and continues with a NaN, so
// the kernel ENDS with NaN results instead of hanging the GPU. The host looks at the word (a plain read of pinned
// memory, no synchronisation) before every single-launch apply and inside mxlo_ctx_sync: fused_fault_check() then
// re-arms the slots, switches the single-launch forms of the ctx off and returns MXLO_EHIP naming what happened.
constexpr unsigned kFaultHouseholder = 1u, kFaultQn = 2u, kFaultHermitian = 4u, kFaultKron = 8u;
constexpr unsigned long long kCanonicalNaN = 0x7FF8000000000000ull;
int32_t fused_fault_check(mxlo_ctx *ctx);           // api_ctx.hip
inline unsigned long long fused_timeout_ticks(const mxlo_ctx *ctx) {   // wall_clock64() ticks: the device's constant-rate clock
  return (unsigned long long)ctx->tune.fused_timeout_ms * (unsigned long long)ctx->wall_clock_khz;
}
#if defined(__HIPCC__)
__device__ __forceinline__ unsigned long long poll_slot(const unsigned long long *slot, unsigned long long ticks,
                                                        unsigned *fault, unsigned code) {
  unsigned long long bits, t0 = 0;
  unsigned it = 0;
  while ((bits = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == kSlotEmpty) {
    __builtin_amdgcn_s_sleep(1);
    if ((++it & 255u) == 0) {                       // the clock is read once per 256 polls (each poll is a memory round trip)
      const unsigned long long now = (unsigned long long)wall_clock64();
      if (t0 == 0) t0 = now;
      else if (now - t0 > ticks) {
        __hip_atomic_store(fault, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return kCanonicalNaN;
      }
    }
  }
  return bits;
}
#endif
// true when `grid` workgroups of `kernel` (BLOCK threads, `lds_worst` bytes of dynamic LDS) fit on the device at once.
// Evaluated once per kernel instantiation and device (the occupancy query costs microseconds) — so `lds_worst` must be
// the LARGEST dynamic LDS size the kernel is ever launched with, not the size of the launch at hand: the cached answer
// then holds for every launch (a smaller request can only fit better).
template <auto Kernel, int BLOCK = kBlock>
inline bool coresident(mxlo_ctx *ctx, int64_t grid, size_t lds_worst = 0) {
  static std::atomic<int> cache[64];                // per kernel: blocks per CU + 1 by device ordinal, 0 = unknown
  const int d = ctx->device & 63;
  int per = cache[d].load(std::memory_order_relaxed);
  if (per == 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, Kernel, BLOCK, lds_worst) != hipSuccess) {
      (void)hipGetLastError();
      nb = 0;
    }
    per = nb + 1;
    cache[d].store(per, std::memory_order_relaxed);
  }
  return (int64_t)(per - 1) * ctx->num_cu >= grid;
}

// ---- 16-byte vector types --------------------------------------------------
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Complex element, laid out as Julia's Complex{R} / C's R _Complex: (re, im) adjacent.
template <typename R>
struct alignas(2 * sizeof(R)) cx {
  R re, im;
  __host__ __device__ constexpr cx(R r = R(0), R i = R(0)) : re(r), im(i) {}
};
template <typename T>
struct is_cx : std::false_type {};
template <typename R>
struct is_cx<cx<R>> : std::true_type {};
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <typename T>
struct Vec16;
template <>
struct Vec16<cx<double>> {   // one ComplexF64 per 16-byte access
  using type = f64x2;
  static constexpr int N = 1;
};
template <>
struct Vec16<cx<float>> {    // two ComplexF32
  using type = f32x4;
  static constexpr int N = 2;
};
template <>
struct Vec16<double> {
  using type = f64x2;
  static constexpr int N = 2;
};
template <>
struct Vec16<float> {
  using type = f32x4;
  static constexpr int N = 4;
};

// ---- wave-wide sum of a double, result in EVERY lane --------------------------------------------------------------
// The coefficient recurrences of the quasi-Newton operators (one wave, O(m) .. O(m^2) dependent reductions) spent their
// time in xor-shuffle trees: a 64-bit __shfl_xor is two ds_bpermute_b32 (~100+ cycles of latency each), six dependent
// levels per sum. DPP moves run on the VALU at register speed: quad_perm x2, row_shr:4, row_shr:8, then row_bcast:15 /
// row_bcast:31 (GFX9-family wave64 modes) leave the total in lane 63, which v_readlane broadcasts. Deterministic (one
// fixed tree), ~20x shorter than the shuffle tree. Masked-off / out-of-row sources read 0 (bound_ctrl).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_shift_or_zero(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
  return __hiloint2double(hi, lo);
#else
  return v;   // host pass of the single-source compile: never executed
#endif
}
__device__ __forceinline__ double wave_allsum(double v) {
  v += dpp_shift_or_zero<0xb1, 0xf>(v);    // quad_perm [1,0,3,2]
  v += dpp_shift_or_zero<0x4e, 0xf>(v);    // quad_perm [2,3,0,1]: every lane holds its quad's sum
  v += dpp_shift_or_zero<0x114, 0xf>(v);   // row_shr:4
  v += dpp_shift_or_zero<0x118, 0xf>(v);   // row_shr:8: lanes 12..15 of a row hold the row's sum
  v += dpp_shift_or_zero<0x142, 0xa>(v);   // row_bcast:15 into rows 1, 3
  v += dpp_shift_or_zero<0x143, 0xc>(v);   // row_bcast:31 into rows 2, 3: lane 63 holds the wave's sum
#if defined(__HIP_DEVICE_COMPILE__)
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
#else
  return v;
#endif
}

// ---- halving butterfly steps without LDS ---------------------------------------------------------------------------
// One step of a halving reduction over the lanes: every lane holds TWO partial sums x, y; lanes whose bit B is clear keep
// x and hand y to lane l ^ B, the others keep y and hand x over — afterwards each lane holds ONE sum over twice the lanes:
//   bit clear: x[l] + x[l ^ B]          bit set: y[l] + y[l ^ B]
// i.e. exactly  (bit ? y : x) + __shfl_xor(bit ? x : y, B)  (the adds are the same two operands: same bits), but a 64-bit
// __shfl_xor is two ds_bpermute_b32 round trips through LDS. B = 32 / 16 are ONE v_permlane32_swap / v_permlane16_swap per
// register half (gfx950: the swap IS the exchange — no select at all), B = 8 / 4 DPP row moves under bank masks.
__device__ __forceinline__ double halve_step32(double x, double y) {
#if defined(__HIP_DEVICE_COMPILE__)
  const auto l = __builtin_amdgcn_permlane32_swap(__double2loint(x), __double2loint(y), false, false);
  const auto h = __builtin_amdgcn_permlane32_swap(__double2hiint(x), __double2hiint(y), false, false);
  // [0]: lanes 0..31 x[l], lanes 32..63 y[l - 32];  [1]: lanes 0..31 x[l + 32], lanes 32..63 y[l]
  return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
#else
  return x + y;
#endif
}
__device__ __forceinline__ double halve_step16(double x, double y) {
#if defined(__HIP_DEVICE_COMPILE__)
  const auto l = __builtin_amdgcn_permlane16_swap(__double2loint(x), __double2loint(y), false, false);
  const auto h = __builtin_amdgcn_permlane16_swap(__double2hiint(x), __double2hiint(y), false, false);
  // [0]: even rows x[l], odd rows y[l - 16];  [1]: even rows x[l + 16], odd rows y[l]
  return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
#else
  return x + y;
#endif
}
// CTRL_UP: lane l reads lane l + B of its row (row_shl:B), CTRL_DN: lane l - B (row_shr:B); BANKS_CLR = the banks
// (4-lane groups of a row) whose lanes have bit B clear.
template <int CTRL_UP, int CTRL_DN, int BANKS_CLR>
__device__ __forceinline__ double halve_step_dpp(double x, double y) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BANKS_SET = 0xf & ~BANKS_CLR;
  int kl = __double2loint(x), kh = __double2hiint(x);       // keep: x where the bit is clear, y where it is set
  kl = __builtin_amdgcn_update_dpp(kl, __double2loint(y), 0xe4, 0xf, BANKS_SET, false);   // quad_perm [0,1,2,3]
  kh = __builtin_amdgcn_update_dpp(kh, __double2hiint(y), 0xe4, 0xf, BANKS_SET, false);
  int rl = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL_UP, 0xf, BANKS_CLR, false);   // x[l + B] into the clear lanes
  int rh = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL_UP, 0xf, BANKS_CLR, false);
  rl = __builtin_amdgcn_update_dpp(rl, __double2loint(y), CTRL_DN, 0xf, BANKS_SET, false);      // y[l - B] into the set lanes
  rh = __builtin_amdgcn_update_dpp(rh, __double2hiint(y), CTRL_DN, 0xf, BANKS_SET, false);
  return __hiloint2double(kh, kl) + __hiloint2double(rh, rl);
#else
  return x + y;
#endif
}
__device__ __forceinline__ double halve_step8(double x, double y) { return halve_step_dpp<0x108, 0x118, 0x3>(x, y); }
__device__ __forceinline__ double halve_step4(double x, double y) { return halve_step_dpp<0x104, 0x114, 0x5>(x, y); }
// v[l] + v[l ^ 2], v[l] + v[l ^ 1] in every lane (quad permutes)
__device__ __forceinline__ double pair_step2(double v) { return v + dpp_shift_or_zero<0x4e, 0xf>(v); }
__device__ __forceinline__ double pair_step1(double v) { return v + dpp_shift_or_zero<0xb1, 0xf>(v); }

// ---- alignment analysis for the vector path ---------------------------------
// All operands of an elementwise kernel can use 16-byte accesses iff they share
// the same address modulo 16; then `head` leading elements are peeled so the
// body starts 16-byte aligned. Returns -1 when the operands disagree.
template <typename T>
inline int64_t common_head(std::initializer_list<const void *> ptrs) {
  int64_t mis = -1;
  for (const void *p : ptrs) {
    if (!p) continue;
    int64_t m = (int64_t)((uintptr_t)p & 15u);
    if (m % (int64_t)sizeof(T)) return -1;
    if (mis < 0) mis = m;
    else if (mis != m) return -1;
  }
  if (mis <= 0) return 0;
  return (16 - mis) / (int64_t)sizeof(T);
}

inline int grid_for(const mxlo_ctx *ctx, int64_t work_items, int64_t items_per_block, int per_cu) {
  int64_t need = (work_items + items_per_block - 1) / items_per_block;
  int64_t g = need;
  if (per_cu > 0) {  // per_cu == 0: one work chunk per workgroup (no persistent loop)
    const int64_t cap = (int64_t)ctx->num_cu * per_cu;
    if (g > cap) g = cap;
  }
  if (g > 0x7fffffffLL) g = 0x7fffffffLL;
  return (int)(g < 1 ? 1 : g);
}

// qn.hip: current generation of a live quasi-Newton handle (false: the handle was destroyed)
bool qn_generation(const mxlo_qn *h, int64_t *gen);

// reductions.hip
int32_t finalize_and_reduce(mxlo_ctx *ctx, int ncols, int nblocks, double *out_dev);
// ... also posting the results (and extra_n earlier doubles) to mapped pinned host memory, reductions.hip
int32_t finalize_and_post(mxlo_ctx *ctx, int ncols, int nblocks, double *out_dev, double *post_dev, unsigned long long seq,
                          const double *extra_src, int extra_n);
int32_t allreduce_hook(mxlo_ctx *ctx, double *dev, int64_t count);
template <typename T>
int32_t panel_dots(mxlo_ctx *ctx, const T *const *cols_host, int ncols, const T *x, int64_t n,
                   double *out_dev);
// the dots pass alone: per-workgroup partials left in ctx->partials, *nblocks per column (<= 4 columns, n >= 1)
template <typename T>
int32_t panel_dots_partials(mxlo_ctx *ctx, const T *const *cols_host, int ncols, const T *x, int64_t n, int *nblocks);
// out1[c] = dot(cols[c], x1), out2[c] = dot(cols[c], x2) in one pass; all operands are aligned, padded panel columns
template <typename T>
int32_t panel_dots2(mxlo_ctx *ctx, const T *const *cols, int ncols, const T *x1, const T *x2, int64_t n_padded,
                    double *out1, double *out2);
// push! pass (reductions.hip): dual-x dots over <= 10 panel columns with the new pair in registers, optional in-pass
// stores into the slot being replaced; LOCAL sums (the caller runs the all-reduce hook)
// Extra jobs the pass's finalize launch can carry, so that a launch-bound push! needs no separate launches for them:
//   cp_*: cp_dst[i] = cp_src[i], i < cp_n (doubles; cp_src was written by an EARLIER kernel) — the inverse push!'s
//         S'y_new -> SY column copy;
//   post: x1·x2 and x2·x2 also go to mapped pinned host memory as (value, sequence number) pairs post[0..1], post[2..3]
//         — the decision scalars of an L-BFGS push!, which the host polls for (qn.hip: await_posted_pairs).
struct PushExtras {
  const double *cp_src = nullptr;
  double *cp_dst = nullptr;
  int cp_n = 0;
  double *post = nullptr;
  unsigned long long post_seq = 0;
};
template <typename T>
int32_t panel_push_pass(mxlo_ctx *ctx, const T *const *cols, int ncols, int slot, int slot_src, const T *x1,
                        const T *x2, int64_t n, int64_t n_padded, T *st1, T *st2, T *stb, double sq, double *out1,
                        double *out2, double *out_x1x2, double *out_x2x2, double *out_bb, double *out_x1x1 = nullptr,
                        const PushExtras *extras = nullptr);

}  // namespace mxlo
