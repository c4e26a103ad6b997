// api_ctx.hip — context, device-memory helpers, timers, error reporting of libmxlo.so.
#include "common.h"
#include <vector>

namespace mxlo {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace mxlo

using namespace mxlo;

MXLO_API const char *mxlo_version(void) { return "mxlo 0.1.0 (gfx950)"; }

MXLO_API const char *mxlo_status_string(int32_t s) {
  switch (s) {
    case MXLO_OK: return "ok";
    case MXLO_EINVAL: return "invalid argument";
    case MXLO_ESHAPE: return "shape mismatch";
    case MXLO_EHIP: return "HIP runtime error";
    case MXLO_ENOMEM: return "out of device memory";
    case MXLO_ESTATE: return "wrong variant for this operator state";
    case MXLO_EDOMAIN: return "argument outside domain";
    case MXLO_EREDUCE: return "all-reduce hook failed";
    default: return "unknown status";
  }
}

MXLO_API const char *mxlo_last_error(void) { return g_err; }

MXLO_API int32_t mxlo_ctx_create(int32_t device_id, void *stream, mxlo_ctx **out) {
  MXLO_REQUIRE(out != nullptr, MXLO_EINVAL, "mxlo_ctx_create: out is NULL");
  int ndev = 0;
  MXLO_HIP(hipGetDeviceCount(&ndev));
  MXLO_REQUIRE(device_id >= 0 && device_id < ndev, MXLO_EINVAL,
               "mxlo_ctx_create: device %d not in [0,%d)", device_id, ndev);
  DeviceGuard guard(device_id);   // the caller's current device is restored on return
  hipDeviceProp_t prop;
  MXLO_HIP(hipGetDeviceProperties(&prop, device_id));
  mxlo_ctx *ctx = new mxlo_ctx();
  ctx->device = device_id;
  ctx->stream = (hipStream_t)stream;
  ctx->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  {
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device_id) == hipSuccess && khz > 0) ctx->wall_clock_khz = khz;
    else (void)hipGetLastError();
  }
  hipError_t e = hipMalloc((void **)&ctx->partials, sizeof(double) * kMaxRedCols * kMaxRedBlocks);
  if (e == hipSuccess) e = hipMalloc((void **)&ctx->scalars, sizeof(double) * kScalarSlots);
  if (e == hipSuccess) e = hipMalloc((void **)&ctx->ticket, 64);
  if (e == hipSuccess) e = hipMalloc((void **)&ctx->xslots, sizeof(unsigned long long) * (2 * kFusedSlots + 8));
  if (e != hipSuccess) {
    set_error("mxlo_ctx_create: workspace allocation failed: %s", hipGetErrorString(e));
    if (ctx->partials) (void)hipFree(ctx->partials);
    if (ctx->scalars) (void)hipFree(ctx->scalars);
    if (ctx->ticket) (void)hipFree(ctx->ticket);
    delete ctx;
    return MXLO_ENOMEM;
  }
  MXLO_HIP(hipMemsetAsync(ctx->scalars, 0, sizeof(double) * kScalarSlots, ctx->stream));
  MXLO_HIP(hipMemsetAsync(ctx->ticket, 0, 64, ctx->stream));
  {  // every exchange slot empty, epoch 0
    std::vector<unsigned long long> init(2 * kFusedSlots + 8, kSlotEmpty);
    for (int i = 0; i < 8; ++i) init[2 * kFusedSlots + i] = 0;
    MXLO_HIP(hipMemcpy(ctx->xslots, init.data(), init.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
  }
  {  // the same for the single-launch quasi-Newton apply: 2 sets of 40 columns x 256 workgroups
    constexpr size_t kQ = 2 * kQnfSlots + 8;
    if (hipMalloc((void **)&ctx->qslots, sizeof(unsigned long long) * kQ) == hipSuccess) {
      std::vector<unsigned long long> init(kQ, kSlotEmpty);
      for (int i = 0; i < 8; ++i) init[kQ - 8 + i] = 0;
      MXLO_HIP(hipMemcpy(ctx->qslots, init.data(), kQ * sizeof(unsigned long long), hipMemcpyHostToDevice));
    } else {
      ctx->qslots = nullptr;     // the four-launch apply is used instead
      (void)hipGetLastError();
    }
  }
  {  // fault word of the single-launch kernels: pinned + device-mapped, so that the host reads it without a sync
    void *hp = nullptr, *dp = nullptr;
    if (hipHostMalloc(&hp, 64, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) {
      memset(hp, 0, 64);
      ctx->fault_host = (unsigned *)hp;
      ctx->fault_dev = (unsigned *)dp;
    } else {                       // no mapped host memory: the single-launch forms stay off (nothing could report a timeout)
      if (hp) (void)hipHostFree(hp);
      (void)hipGetLastError();
      ctx->tune.house_fused = 0;
      ctx->tune.qn_fused_small = 0;
      ctx->tune.qn_persist = 0;
    }
  }
  *out = ctx;
  return MXLO_OK;
}

namespace mxlo {
static int32_t rearm_fused_slots(mxlo_ctx *ctx) {
  if (ctx->xslots) {
    std::vector<unsigned long long> init(2 * kFusedSlots + 8, kSlotEmpty);
    for (int i = 0; i < 8; ++i) init[2 * kFusedSlots + i] = 0;
    MXLO_HIP(hipMemcpy(ctx->xslots, init.data(), init.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
  }
  if (ctx->qslots) {
    constexpr size_t kQ = 2 * kQnfSlots + 8;
    std::vector<unsigned long long> init(kQ, kSlotEmpty);
    for (int i = 0; i < 8; ++i) init[kQ - 8 + i] = 0;
    MXLO_HIP(hipMemcpy(ctx->qslots, init.data(), kQ * sizeof(unsigned long long), hipMemcpyHostToDevice));
  }
  return MXLO_OK;
}

int32_t fused_fault_check(mxlo_ctx *ctx) {
  if (!ctx->fault_host || ctx->capturing) return MXLO_OK;
  const unsigned code = __atomic_load_n(ctx->fault_host, __ATOMIC_RELAXED);
  if (code == 0) return MXLO_OK;
  // A single-launch apply gave up waiting for its peer workgroups. Its output is NaN; the exchange slots and the epoch
  // word are in an unknown state. Drain the stream, re-arm everything, and keep the single-launch forms off: whatever
  // kept the grid from being co-resident is likely to still be there.
  (void)hipStreamSynchronize(ctx->stream);
  ctx->tune.house_fused = 0;
  ctx->tune.qn_fused_small = 0;
  ctx->tune.qn_persist = 0;
  ctx->tune.herm_single = 0;
  ctx->tune.kron_fuse = 0;
  if (ctx->kron_cnt) (void)hipMemset(ctx->kron_cnt, 0, sizeof(unsigned) * (size_t)ctx->kron_cnt_n);
  ctx->herm_slots_dirty = true;
  __atomic_store_n(ctx->fault_host, 0u, __ATOMIC_RELAXED);
  (void)rearm_fused_slots(ctx);
  set_error("a single-launch %s apply on this ctx timed out after %d ms waiting for its peer workgroups (the launch was "
            "not fully co-resident: GPU shared with other processes, CU masking, or a killed launch left the exchange "
            "slots inconsistent); that apply stored NaN. The exchange state has been re-armed and the single-launch "
            "forms (house_fused, qn_fused_small, qn_persist, herm_single, kron_fuse) are now OFF for this ctx — repeat the apply",
            code == kFaultHouseholder ? "opHouseholder" : (code == kFaultHermitian ? "opHermitian" : (code == kFaultKron ? "kron" : "quasi-Newton")), ctx->tune.fused_timeout_ms);
  return MXLO_EHIP;
}
}  // namespace mxlo

MXLO_API int32_t mxlo_ctx_destroy(mxlo_ctx *ctx) {
  if (!ctx) return MXLO_OK;
  MXLO_DEVICE_GUARD(ctx);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->partials) (void)hipFree(ctx->partials);
  if (ctx->scalars) (void)hipFree(ctx->scalars);
  if (ctx->ticket) (void)hipFree(ctx->ticket);
  if (ctx->xslots) (void)hipFree(ctx->xslots);
  if (ctx->qslots) (void)hipFree(ctx->qslots);
  if (ctx->herm_slots) (void)hipFree(ctx->herm_slots);
  if (ctx->kron_cnt) (void)hipFree(ctx->kron_cnt);
  if (ctx->fault_host) (void)hipHostFree(ctx->fault_host);
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  if (ctx->switch_event) (void)hipEventDestroy(ctx->switch_event);
  delete ctx;
  return MXLO_OK;
}

MXLO_API int32_t mxlo_ctx_set_stream(mxlo_ctx *ctx, void *stream) {
  MXLO_REQUIRE(ctx, MXLO_EINVAL, "ctx is NULL");
  MXLO_DEVICE_GUARD(ctx);
  hipStream_t next = (hipStream_t)stream;
  MXLO_REQUIRE(!(ctx->capturing && next != ctx->stream), MXLO_ESTATE,
               "mxlo_ctx_set_stream: a graph capture is open on this ctx; end it before changing the stream");
  if (next != ctx->stream) {
    // the reduction workspace, the scalar buffer and the quasi-Newton handles of this ctx are ordered by the
    // stream: work already queued on the old stream must complete before the new stream touches them
    if (!ctx->switch_event) MXLO_HIP(hipEventCreateWithFlags(&ctx->switch_event, hipEventDisableTiming));
    MXLO_HIP(hipEventRecord(ctx->switch_event, ctx->stream));
    MXLO_HIP(hipStreamWaitEvent(next, ctx->switch_event, 0));
  }
  ctx->stream = next;
  return MXLO_OK;
}

MXLO_API int32_t mxlo_ctx_create_stream(mxlo_ctx *ctx, void **out) {
  MXLO_REQUIRE(ctx, MXLO_EINVAL, "ctx is NULL");
  MXLO_DEVICE_GUARD(ctx);
  if (!ctx->own_stream) {
    MXLO_HIP(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
  }
  MXLO_TRY(mxlo_ctx_set_stream(ctx, (void *)ctx->own_stream));
  if (out) *out = (void *)ctx->own_stream;
  return MXLO_OK;
}

// ---- hipGraph capture of launch-bound sequences -------------------------------------------------
struct mxlo_graph {
  mxlo_ctx *ctx = nullptr;
  hipStream_t stream = nullptr;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  // state baked into the recorded kernel arguments (slot order, γ, active set, which kernels run; the opHermitian
  // workspace pointer): a replay after push!/reset!/a workspace reallocation would silently use stale values
  std::vector<std::pair<mxlo_qn *, int64_t>> qn;
  int64_t scratch_generation = -1;   // -1: no opHermitian apply inside the graph
  // Short linear chains of kernel / memset nodes are replayed by re-issuing the recorded launches one by one: on this
  // runtime hipGraphLaunch costs ~9 us of host time plus ~3 us per node, a plain launch with the node's own
  // (already marshalled) argument block ~2 us — so for the launch-bound sequences graphs exist for, the direct
  // chain is never slower than the eager calls, while hipGraphLaunch of a 1-3 node graph is.
  struct Step {
    bool is_memset = false;
    hipKernelNodeParams k{};
    hipMemsetParams m{};
  };
  std::vector<Step> chain;           // empty: replay through hipGraphLaunch
};

// Walks the captured graph; fills g->chain when it is one dependency chain of kernel / 1-D memset nodes.
static void try_linearise(mxlo_graph *g, int max_nodes) {
  size_t nn = 0, nr = 0;
  if (hipGraphGetNodes(g->graph, nullptr, &nn) != hipSuccess || nn == 0 || (int)nn > max_nodes) return;
  if (hipGraphGetRootNodes(g->graph, nullptr, &nr) != hipSuccess || nr != 1) return;
  hipGraphNode_t cur = nullptr;
  if (hipGraphGetRootNodes(g->graph, &cur, &nr) != hipSuccess) return;
  std::vector<mxlo_graph::Step> chain;
  while (true) {
    hipGraphNodeType ty;
    if (hipGraphNodeGetType(cur, &ty) != hipSuccess) return;
    mxlo_graph::Step st;
    if (ty == hipGraphNodeTypeKernel) {
      if (hipGraphKernelNodeGetParams(cur, &st.k) != hipSuccess || !st.k.func || !st.k.kernelParams || st.k.extra) return;
    } else if (ty == hipGraphNodeTypeMemset) {
      st.is_memset = true;
      if (hipGraphMemsetNodeGetParams(cur, &st.m) != hipSuccess || st.m.height != 1) return;
      if (st.m.elementSize != 1 && st.m.elementSize != 2 && st.m.elementSize != 4) return;
    } else {
      return;
    }
    chain.push_back(st);
    size_t nd = 0;
    if (hipGraphNodeGetDependentNodes(cur, nullptr, &nd) != hipSuccess || nd > 1) return;
    if (nd == 0) break;
    if (chain.size() > nn) return;
    hipGraphNode_t next = nullptr;
    if (hipGraphNodeGetDependentNodes(cur, &next, &nd) != hipSuccess || !next) return;
    cur = next;
  }
  if (chain.size() == nn) g->chain = std::move(chain);
  (void)hipGetLastError();
}

MXLO_API int32_t mxlo_graph_begin(mxlo_ctx *ctx) {
  MXLO_REQUIRE(ctx, MXLO_EINVAL, "ctx is NULL");
  MXLO_DEVICE_GUARD(ctx);
  MXLO_REQUIRE(!ctx->capturing, MXLO_ESTATE, "mxlo_graph_begin: a capture is already open on this ctx");
  MXLO_REQUIRE(ctx->stream != nullptr, MXLO_ESTATE,
               "mxlo_graph_begin: the default stream cannot be captured; give the ctx a stream "
               "(mxlo_ctx_set_stream / mxlo_ctx_create_stream)");
  MXLO_HIP(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
  ctx->capturing = true;
  ctx->captured_qn.clear();
  ctx->scratch_used_in_capture = false;
  return MXLO_OK;
}

MXLO_API int32_t mxlo_graph_end(mxlo_ctx *ctx, mxlo_graph **out) {
  MXLO_REQUIRE(ctx && out, MXLO_EINVAL, "mxlo_graph_end: NULL argument");
  MXLO_DEVICE_GUARD(ctx);
  MXLO_REQUIRE(ctx->capturing, MXLO_ESTATE, "mxlo_graph_end: no capture is open on this ctx");
  ctx->capturing = false;
  hipGraph_t graph = nullptr;
  hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
  if (e != hipSuccess || !graph) {  // a call inside the capture was not capturable (sync, allocation)
    set_error("mxlo_graph_end: capture failed: %s — only stream-ordered work (mul!, solve_shifted_system!, "
              "diag!) can be captured; push! and first-use workspace growth synchronise", hipGetErrorString(e));
    (void)hipGetLastError();
    return MXLO_EHIP;
  }
  mxlo_graph *g = new mxlo_graph();
  g->ctx = ctx;
  g->stream = ctx->stream;
  g->graph = graph;
  g->qn = ctx->captured_qn;
  g->scratch_generation = ctx->scratch_used_in_capture ? ctx->scratch_generation : -1;
  ctx->captured_qn.clear();
  e = hipGraphInstantiate(&g->exec, graph, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e));
    (void)hipGraphDestroy(graph);
    delete g;
    return MXLO_EHIP;
  }
  if (ctx->tune.graph_direct_max > 0) try_linearise(g, ctx->tune.graph_direct_max);
  *out = g;
  return MXLO_OK;
}

MXLO_API int32_t mxlo_graph_launch(mxlo_graph *g) {
  MXLO_REQUIRE(g && g->exec, MXLO_EINVAL, "mxlo_graph_launch: NULL graph");
  MXLO_DEVICE_GUARD(g->ctx);
  for (const auto &pr : g->qn) {
    int64_t gen = 0;
    MXLO_REQUIRE(qn_generation(pr.first, &gen), MXLO_ESTATE,
                 "mxlo_graph_launch: a quasi-Newton operator applied inside this graph was destroyed");
    MXLO_REQUIRE(gen == pr.second, MXLO_ESTATE,
                 "mxlo_graph_launch: a quasi-Newton operator applied inside this graph changed state (push!/reset!/mode) "
                 "after the capture: slot order, scaling factor and kernel choice are baked into the graph — recapture");
  }
  MXLO_REQUIRE(g->scratch_generation < 0 || g->scratch_generation == g->ctx->scratch_generation, MXLO_ESTATE,
               "mxlo_graph_launch: the opHermitian workspace recorded in this graph was reallocated (a larger n arrived) "
               "— recapture");
  // A graph replays on the stream it was captured from. If the ctx has moved to another stream since, order the two:
  // the workspaces, the scalar buffer and the exchange slots of the single-launch kernels belong to the ctx, not to a
  // stream, so a replay must neither overtake nor be overtaken by the ctx's direct launches.
  mxlo_ctx *ctx = g->ctx;
  const bool foreign = g->stream != ctx->stream;
  struct Rejoin {                      // after the replay: the ctx stream waits for it (also on the error paths below)
    mxlo_ctx *ctx; hipStream_t gs; bool on;
    ~Rejoin() {
      if (on && hipEventRecord(ctx->switch_event, gs) == hipSuccess) (void)hipStreamWaitEvent(ctx->stream, ctx->switch_event, 0);
    }
  } rejoin{ctx, g->stream, false};
  if (foreign) {
    MXLO_REQUIRE(!ctx->capturing, MXLO_ESTATE, "mxlo_graph_launch: a capture is open on this ctx's stream");
    if (!ctx->switch_event) MXLO_HIP(hipEventCreateWithFlags(&ctx->switch_event, hipEventDisableTiming));
    MXLO_HIP(hipEventRecord(ctx->switch_event, ctx->stream));
    MXLO_HIP(hipStreamWaitEvent(g->stream, ctx->switch_event, 0));
    rejoin.on = true;
  }
  if (!g->chain.empty()) {
    for (const auto &st : g->chain) {
      if (st.is_memset) {
        if (st.m.elementSize == 4) MXLO_HIP(hipMemsetD32Async((hipDeviceptr_t)st.m.dst, (int)st.m.value, st.m.width, g->stream));
        else if (st.m.elementSize == 2) MXLO_HIP(hipMemsetD16Async((hipDeviceptr_t)st.m.dst, (unsigned short)st.m.value, st.m.width, g->stream));
        else MXLO_HIP(hipMemsetD8Async((hipDeviceptr_t)st.m.dst, (unsigned char)st.m.value, st.m.width, g->stream));
      } else {
        MXLO_HIP(hipLaunchKernel(st.k.func, st.k.gridDim, st.k.blockDim, st.k.kernelParams, st.k.sharedMemBytes, g->stream));
      }
    }
    return MXLO_OK;
  }
  MXLO_HIP(hipGraphLaunch(g->exec, g->stream));
  return MXLO_OK;
}

MXLO_API int32_t mxlo_graph_info(mxlo_graph *g, int64_t info[2]) {
  MXLO_REQUIRE(g && info, MXLO_EINVAL, "mxlo_graph_info: NULL argument");
  size_t nn = 0;
  MXLO_DEVICE_GUARD(g->ctx);
  MXLO_HIP(hipGraphGetNodes(g->graph, nullptr, &nn));
  info[0] = (int64_t)nn;
  info[1] = g->chain.empty() ? 0 : 1;
  return MXLO_OK;
}

MXLO_API int32_t mxlo_graph_destroy(mxlo_graph *g) {
  if (!g) return MXLO_OK;
  MXLO_DEVICE_GUARD(g->ctx);
  if (g->exec) (void)hipGraphExecDestroy(g->exec);
  if (g->graph) (void)hipGraphDestroy(g->graph);
  delete g;
  return MXLO_OK;
}

MXLO_API int32_t mxlo_debug_counters(int64_t out[12]) {
  MXLO_REQUIRE(out, MXLO_EINVAL, "mxlo_debug_counters: out is NULL");
  ApiCounters &c = api_counters();
  const int64_t v[12] = {c.n_malloc, c.n_free, c.n_h2d, c.n_d2h, c.n_d2d, c.n_d2h_bytes, c.n_stream_sync,
                         c.n_device_sync, c.n_event_sync, c.n_memset_async, c.n_launch, c.n_blocking_copy};
  for (int i = 0; i < 12; ++i) out[i] = v[i];
  return MXLO_OK;
}

MXLO_API int32_t mxlo_ctx_sync(mxlo_ctx *ctx) {
  MXLO_REQUIRE(ctx, MXLO_EINVAL, "ctx is NULL");
  MXLO_DEVICE_GUARD(ctx);
  MXLO_HIP(hipStreamSynchronize(ctx->stream));
  return fused_fault_check(ctx);      // a single-launch apply that timed out is reported here at the latest
}

MXLO_API int32_t mxlo_ctx_info(mxlo_ctx *ctx, int64_t info[4]) {
  MXLO_REQUIRE(ctx && info, MXLO_EINVAL, "ctx/info is NULL");
  info[0] = ctx->device;
  info[1] = ctx->num_cu;
  info[2] = (int64_t)sizeof(double) * (kMaxRedCols * kMaxRedBlocks + kScalarSlots);
  info[3] = kMaxRedCols;
  return MXLO_OK;
}

MXLO_API int32_t mxlo_ctx_tune(mxlo_ctx *ctx, const char *key, int64_t value) {
  MXLO_REQUIRE(ctx && key, MXLO_EINVAL, "ctx/key is NULL");
  if (!strcmp(key, "blocks_per_cu")) {
    MXLO_REQUIRE(value >= 0 && value <= 64, MXLO_EINVAL, "blocks_per_cu out of range");
    ctx->tune.blocks_per_cu = (int)value;
  } else if (!strcmp(key, "nt_min_bytes")) {
    MXLO_REQUIRE(value >= 0, MXLO_EINVAL, "nt_min_bytes out of range");
    ctx->tune.nt_min_bytes = value;
  } else if (!strcmp(key, "red_blocks_per_cu")) {
    MXLO_REQUIRE(value >= 1 && value * ctx->num_cu <= kMaxRedBlocks, MXLO_EINVAL,
                 "red_blocks_per_cu out of range");
    ctx->tune.red_blocks_per_cu = (int)value;
  } else if (!strcmp(key, "graph_direct_max")) {
    MXLO_REQUIRE(value >= 0 && value <= 4096, MXLO_EINVAL, "graph_direct_max must be in 0..4096 (0: always hipGraphLaunch)");
    ctx->tune.graph_direct_max = (int)value;
  } else if (!strcmp(key, "house_fused")) {
    ctx->tune.house_fused = value != 0;
  } else if (!strcmp(key, "cherm_two_pass")) {
    ctx->tune.cherm_two_pass = value != 0;
  } else if (!strcmp(key, "house_inline_n")) {
    MXLO_REQUIRE(value >= 0, MXLO_EINVAL, "house_inline_n out of range");
    ctx->tune.house_inline_n = value;
  } else if (!strcmp(key, "house_reverse")) {
    ctx->tune.house_reverse = value != 0;
  } else if (!strcmp(key, "lbfgs_inv_mode")) {
    MXLO_REQUIRE(value == MXLO_INV_TWOPASS || value == MXLO_INV_REFORDER, MXLO_EINVAL,
                 "lbfgs_inv_mode must be MXLO_INV_TWOPASS or MXLO_INV_REFORDER");
    ctx->tune.lbfgs_inv_mode = (int)value;
  } else if (!strcmp(key, "gemm_tile")) {
    MXLO_REQUIRE(value == 0 || value == 32 || value == 64 || value == 128 || value == -1, MXLO_EINVAL,
                 "gemm_tile must be 0 (auto), 32, 64, 128 or -1 (generic kernel)");
    ctx->tune.gemm_tile = (int)value;
  } else if (!strcmp(key, "extend_tiles_per_block")) {
    MXLO_REQUIRE(value >= 0 && value <= 1024, MXLO_EINVAL, "extend_tiles_per_block must be in 0..1024 (0 = auto)");
    ctx->tune.extend_tiles_per_block = (int)value;
  } else if (!strcmp(key, "fuse_finalize")) {
    MXLO_REQUIRE(value == 0 || value == 1, MXLO_EINVAL, "fuse_finalize must be 0 or 1");
    ctx->tune.fuse_finalize = (int)value;
  } else if (!strcmp(key, "combine_blocks_per_cu")) {
    MXLO_REQUIRE(value >= 0 && value <= 64, MXLO_EINVAL, "combine_blocks_per_cu out of range");
    ctx->tune.combine_blocks_per_cu = (int)value;
  } else if (!strcmp(key, "qn_fused_max_grid")) {
    MXLO_REQUIRE(value >= 1 && value <= kQnfMaxGrid, MXLO_EINVAL, "qn_fused_max_grid must be in 1..256");
    ctx->tune.qn_fused_max_grid = (int)value;
  } else if (!strcmp(key, "qn_fused_batch12")) {
    MXLO_REQUIRE(value == 0 || value == 1, MXLO_EINVAL, "qn_fused_batch12 must be 0 or 1");
    ctx->tune.qn_fused_batch12 = (int)value;
  } else if (!strcmp(key, "qn_fused_small")) {
    MXLO_REQUIRE(value == 0 || value == 1, MXLO_EINVAL, "qn_fused_small must be 0 or 1");
    ctx->tune.qn_fused_small = (int)value;
  } else if (!strcmp(key, "qn_persist")) {
    MXLO_REQUIRE(value == 0 || value == 1, MXLO_EINVAL, "qn_persist must be 0 or 1");
    ctx->tune.qn_persist = (int)value;
  } else if (!strcmp(key, "qn_persist_min_n")) {
    MXLO_REQUIRE(value >= 1, MXLO_EINVAL, "qn_persist_min_n must be >= 1");
    ctx->tune.qn_persist_min_n = value;
  } else if (!strcmp(key, "qn_persist_max_bytes")) {
    MXLO_REQUIRE(value >= 0, MXLO_EINVAL, "qn_persist_max_bytes must be >= 0");
    ctx->tune.qn_persist_max_bytes = value;
  } else if (!strcmp(key, "qn_persist_min_bytes")) {
    MXLO_REQUIRE(value >= 0, MXLO_EINVAL, "qn_persist_min_bytes must be >= 0");
    ctx->tune.qn_persist_min_bytes = value;
  } else if (!strcmp(key, "qn_persist_reverse")) {
    MXLO_REQUIRE(value == 0 || value == 1, MXLO_EINVAL, "qn_persist_reverse must be 0 or 1");
    ctx->tune.qn_persist_reverse = (int)value;
  } else if (!strcmp(key, "qn_persist_prefetch")) {
    MXLO_REQUIRE(value == 0 || value == 1, MXLO_EINVAL, "qn_persist_prefetch must be 0 or 1");
    ctx->tune.qn_persist_prefetch = (int)value;
  } else if (!strcmp(key, "qn_persist_lds")) {
    MXLO_REQUIRE(value == 0 || value == 1, MXLO_EINVAL, "qn_persist_lds must be 0 or 1");
    ctx->tune.qn_persist_lds = (int)value;
  } else if (!strcmp(key, "qn_persist_lds_pad")) {
    MXLO_REQUIRE(value >= 0 && value <= 112 * 1024, MXLO_EINVAL, "qn_persist_lds_pad must be in 0..114688 bytes");
    ctx->tune.qn_persist_lds_pad = (int)value;
  } else if (!strcmp(key, "kron_fuse")) {
    MXLO_REQUIRE(value >= 0 && value <= 2, MXLO_EINVAL, "kron_fuse must be 0, 1 (or 2: timing experiment without the wait, wrong results)");
    ctx->tune.kron_fuse = (int)value;
  } else if (!strcmp(key, "herm_order")) {
    MXLO_REQUIRE(value == 0 || value == 1, MXLO_EINVAL, "herm_order must be 0 or 1");
    ctx->tune.herm_order = (int)value;
  } else if (!strcmp(key, "herm_nt")) {
    MXLO_REQUIRE(value >= -1 && value <= 1, MXLO_EINVAL, "herm_nt must be -1 (by size), 0 or 1");
    ctx->tune.herm_nt = (int)value;
  } else if (!strcmp(key, "herm_dp_min_bytes")) {
    MXLO_REQUIRE(value >= 0, MXLO_EINVAL, "herm_dp_min_bytes must be >= 0");
    ctx->tune.herm_dp_min_bytes = value;
  } else if (!strcmp(key, "herm_nt_min_bytes")) {
    MXLO_REQUIRE(value >= 0, MXLO_EINVAL, "herm_nt_min_bytes must be >= 0");
    ctx->tune.herm_nt_min_bytes = value;
  } else if (!strcmp(key, "house_fused_per_cu")) {
    MXLO_REQUIRE(value == 1 || value == 2, MXLO_EINVAL, "house_fused_per_cu must be 1 or 2");
    ctx->tune.house_fused_per_cu = (int)value;
  } else if (!strcmp(key, "herm_poll_sleep")) {
    MXLO_REQUIRE(value >= 1 && value <= 1024, MXLO_EINVAL, "herm_poll_sleep must be in 1..1024");
    ctx->tune.herm_poll_sleep = (int)value;
  } else if (!strcmp(key, "herm_strip")) {
    MXLO_REQUIRE(value == 0 || value == 1 || value == 2 || value == 8, MXLO_EINVAL, "herm_strip must be 0 (by size), 1, 2 or 8");
    ctx->tune.herm_strip = (int)value;
  } else if (!strcmp(key, "herm_lds_pad")) {
    MXLO_REQUIRE(value >= 0 && value <= 48 * 1024, MXLO_EINVAL, "herm_lds_pad must be in 0..49152 bytes");
    ctx->tune.herm_lds_pad = (int)value;
  } else if (!strcmp(key, "herm_single")) {
    MXLO_REQUIRE(value == 0 || value == 1, MXLO_EINVAL, "herm_single must be 0 or 1");
    ctx->tune.herm_single = (int)value;
  } else if (!strcmp(key, "herm_single_max_bytes")) {
    MXLO_REQUIRE(value >= 0, MXLO_EINVAL, "herm_single_max_bytes must be >= 0");
    ctx->tune.herm_single_max_bytes = value;
  } else if (!strcmp(key, "herm_single_max_n")) {
    MXLO_REQUIRE(value >= 0, MXLO_EINVAL, "herm_single_max_n must be >= 0");
    ctx->tune.herm_single_max_n = value;
  } else if (!strcmp(key, "gemv_n_rows")) {
    MXLO_REQUIRE(value >= 0 && value <= 128, MXLO_EINVAL, "gemv_n_rows must be 0 (off), 1 (auto) or a band height");
    ctx->tune.gemv_n_rows = (int)value;
  } else if (!strcmp(key, "gemvb_n_rows")) {
    MXLO_REQUIRE(value == 0 || value == 1, MXLO_EINVAL, "gemvb_n_rows must be 0 or 1");
    ctx->tune.gemvb_n_rows = (int)value;
  } else if (!strcmp(key, "gemvb_t_lds")) {
    MXLO_REQUIRE(value == 0 || value == 1, MXLO_EINVAL, "gemvb_t_lds must be 0 or 1");
    ctx->tune.gemvb_t_lds = (int)value;
  } else if (!strcmp(key, "combine_reverse")) {
    MXLO_REQUIRE(value == 0 || value == 1, MXLO_EINVAL, "combine_reverse must be 0 or 1");
    ctx->tune.combine_reverse = (int)value;
  } else if (!strcmp(key, "sp_xcds")) {
    MXLO_REQUIRE(value >= 1 && value <= 64, MXLO_EINVAL, "sp_xcds must be in 1..64");
    ctx->tune.sp_xcds = (int)value;
  } else if (!strcmp(key, "fused_timeout_ms")) {
    MXLO_REQUIRE(value >= 1 && value <= 600000, MXLO_EINVAL, "fused_timeout_ms must be in 1..600000");
    ctx->tune.fused_timeout_ms = (int)value;
  } else if (!strcmp(key, "fused_debug_drop")) {
    MXLO_REQUIRE(value >= -1 && value < 4096, MXLO_EINVAL, "fused_debug_drop must be -1 (off) or a workgroup index");
    ctx->tune.fused_debug_drop = (int)value;
  } else if (!strcmp(key, "push_wide")) {
    MXLO_REQUIRE(value == 0 || value == 1, MXLO_EINVAL, "push_wide must be 0 or 1");
    ctx->tune.push_wide = (int)value;
  } else if (!strcmp(key, "push_posted")) {
    MXLO_REQUIRE(value >= 0 && value <= 2, MXLO_EINVAL, "push_posted must be 0, 1 or 2 (2: debug, the posting is treated as lost)");
    ctx->tune.push_posted = (int)value;
  } else if (!strcmp(key, "push_fused")) {
    MXLO_REQUIRE(value == 0 || value == 1, MXLO_EINVAL, "push_fused must be 0 or 1");
    ctx->tune.push_fused = (int)value;
  } else if (!strcmp(key, "dots_max_nc")) {
    MXLO_REQUIRE(value >= 1 && value <= 20, MXLO_EINVAL, "dots_max_nc out of range");
    ctx->tune.dots_max_nc = (int)value;
  } else {
    set_error("mxlo_ctx_tune: unknown key '%s'", key);
    return MXLO_EINVAL;
  }
  return MXLO_OK;
}

MXLO_API int32_t mxlo_ctx_set_allreduce(mxlo_ctx *ctx, mxlo_allreduce_fn fn, void *user) {
  MXLO_REQUIRE(ctx, MXLO_EINVAL, "ctx is NULL");
  ctx->allreduce = fn;
  ctx->allreduce_user = user;
  return MXLO_OK;
}

// ---- memory helpers -----------------------------------------------------------
MXLO_API int32_t mxlo_malloc(mxlo_ctx *ctx, int64_t bytes, void **out) {
  MXLO_REQUIRE(ctx && out && bytes >= 0, MXLO_EINVAL, "mxlo_malloc: bad argument");
  MXLO_DEVICE_GUARD(ctx);
  *out = nullptr;
  if (bytes == 0) return MXLO_OK;
  hipError_t e = hipMalloc(out, (size_t)bytes);
  if (e != hipSuccess) {
    set_error("mxlo_malloc(%lld): %s", (long long)bytes, hipGetErrorString(e));
    return MXLO_ENOMEM;
  }
  return MXLO_OK;
}

MXLO_API int32_t mxlo_free(mxlo_ctx *ctx, void *p) {
  MXLO_REQUIRE(ctx, MXLO_EINVAL, "ctx is NULL");
  MXLO_DEVICE_GUARD(ctx);
  if (!p) return MXLO_OK;
  MXLO_HIP(hipStreamSynchronize(ctx->stream));
  MXLO_HIP(hipFree(p));
  return MXLO_OK;
}

MXLO_API int32_t mxlo_memcpy_h2d(mxlo_ctx *ctx, void *dst, const void *src, int64_t bytes) {
  MXLO_REQUIRE(ctx && bytes >= 0, MXLO_EINVAL, "bad argument");
  MXLO_DEVICE_GUARD(ctx);
  if (!bytes) return MXLO_OK;
  MXLO_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, ctx->stream));
  MXLO_HIP(hipStreamSynchronize(ctx->stream));  // the host buffer may be pageable / reused
  return MXLO_OK;
}

MXLO_API int32_t mxlo_memcpy_d2h(mxlo_ctx *ctx, void *dst, const void *src, int64_t bytes) {
  MXLO_REQUIRE(ctx && bytes >= 0, MXLO_EINVAL, "bad argument");
  MXLO_DEVICE_GUARD(ctx);
  if (!bytes) return MXLO_OK;
  MXLO_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, ctx->stream));
  MXLO_HIP(hipStreamSynchronize(ctx->stream));
  return MXLO_OK;
}

MXLO_API int32_t mxlo_memcpy_d2d(mxlo_ctx *ctx, void *dst, const void *src, int64_t bytes) {
  MXLO_REQUIRE(ctx && bytes >= 0, MXLO_EINVAL, "bad argument");
  MXLO_DEVICE_GUARD(ctx);
  if (!bytes) return MXLO_OK;
  MXLO_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, ctx->stream));
  return MXLO_OK;
}

MXLO_API int32_t mxlo_memset(mxlo_ctx *ctx, void *p, int32_t byte, int64_t bytes) {
  MXLO_REQUIRE(ctx && bytes >= 0, MXLO_EINVAL, "bad argument");
  MXLO_DEVICE_GUARD(ctx);
  if (!bytes) return MXLO_OK;
  MXLO_HIP(hipMemsetAsync(p, byte, (size_t)bytes, ctx->stream));
  return MXLO_OK;
}

// ---- timers ---------------------------------------------------------------------
struct mxlo_timer {
  mxlo_ctx *ctx;
  hipEvent_t e0, e1;
};

MXLO_API int32_t mxlo_timer_create(mxlo_ctx *ctx, mxlo_timer **out) {
  MXLO_REQUIRE(ctx && out, MXLO_EINVAL, "bad argument");
  MXLO_DEVICE_GUARD(ctx);
  mxlo_timer *t = new mxlo_timer();
  t->ctx = ctx;
  MXLO_HIP(hipEventCreate(&t->e0));
  MXLO_HIP(hipEventCreate(&t->e1));
  *out = t;
  return MXLO_OK;
}
MXLO_API int32_t mxlo_timer_start(mxlo_timer *t) {
  MXLO_REQUIRE(t, MXLO_EINVAL, "timer is NULL");
  MXLO_DEVICE_GUARD(t->ctx);
  MXLO_HIP(hipEventRecord(t->e0, t->ctx->stream));
  return MXLO_OK;
}
MXLO_API int32_t mxlo_timer_stop(mxlo_timer *t) {
  MXLO_REQUIRE(t, MXLO_EINVAL, "timer is NULL");
  MXLO_DEVICE_GUARD(t->ctx);
  MXLO_HIP(hipEventRecord(t->e1, t->ctx->stream));
  return MXLO_OK;
}
MXLO_API int32_t mxlo_timer_elapsed_ms(mxlo_timer *t, double *ms) {
  MXLO_REQUIRE(t && ms, MXLO_EINVAL, "bad argument");
  MXLO_DEVICE_GUARD(t->ctx);
  MXLO_HIP(hipEventSynchronize(t->e1));
  float f = 0.f;
  MXLO_HIP(hipEventElapsedTime(&f, t->e0, t->e1));
  *ms = (double)f;
  return MXLO_OK;
}
MXLO_API int32_t mxlo_timer_destroy(mxlo_timer *t) {
  if (!t) return MXLO_OK;
  MXLO_DEVICE_GUARD(t->ctx);
  (void)hipEventDestroy(t->e0);
  (void)hipEventDestroy(t->e1);
  delete t;
  return MXLO_OK;
}
