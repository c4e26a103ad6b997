/* mxlo.h — C ABI of libmxlo.so: the MI355X (gfx950) implementation of the
 * LinearOperators.jl in-place 5-arg `mul!(res, op, v, alpha, beta)` hot path.
 *
 * Every entry point is what the body of ONE reference closure (`prod!`,
 * `tprod!`, `ctprod!`) — or one quasi-Newton state method — would `ccall`.
 * Each declaration cites the reference code it replaces as
 * `src/<file>.jl:<lines>` (paths relative to LinearOperators.jl v2.14.2).
 *
 * Conventions
 *  - plain C, no torch / Julia types: device pointers are `void*`, sizes are
 *    `int64_t`, scalars are `double`.
 *  - every function returns an `int32_t` status (`MXLO_OK` == 0). No exception
 *    ever crosses this boundary; `mxlo_last_error()` returns a thread-local
 *    human-readable message for the last non-zero status.
 *  - all work is enqueued on the ctx's HIP stream and is stream-ordered; the
 *    host returns immediately (`mxlo_ctx_sync` / D2H copies synchronise).
 *  - vectors are dense, contiguous, element-aligned (NOT necessarily 16-byte
 *    aligned: `view(res, k+1:k+m)` of cat / block-diag arrive as base+offset,
 *    src/cat.jl:17-18, src/special-operators.jl:263); matrices column-major
 *    with a leading dimension, exactly Julia's layout.
 *  - `dtype`: MXLO_F64 or MXLO_F32 (element type of res / v / operator data); the `_c` entry points take
 *    MXLO_C64 / MXLO_C32 (ComplexF64 / ComplexF32) for the elementwise leaves and opHouseholder.
 *  - `alpha`, `beta` always arrive as double. Julia does not convert caller
 *    scalars to the element type, so for MXLO_F32 data each scalar carries its
 *    own width flag: MXLO_ALPHA_F64 / MXLO_BETA_F64 say that scalar was a
 *    Float64 in the caller. The alpha-term of `α .* d .* v .+ β .* res` is then
 *    evaluated in promote_type(typeof(α), Float32), the beta-term in
 *    promote_type(typeof(β), Float32), their sum in the wider of the two, and
 *    the result is rounded once on store — all four combinations bit-exact
 *    (`mul!(res32, op32, v32, 2.0, 3.0)`: both flags; the 3-arg `mul!` uses
 *    one(Float32), zero(Float32): no flag, pure fp32, src/operations.jl:38-40).
 *    A scalar without its flag is first rounded to float.
 *  - `beta == 0` means OVERWRITE: `res` is never read (it may hold NaN/Inf
 *    garbage from `similar`, src/operations.jl:45, src/constructors.jl:63-78).
 *  - elementwise arithmetic is evaluated in the reference's association
 *    order with NO fused multiply-add (the library is built with
 *    -ffp-contract=off), so the elementwise leaves are bit-identical to the
 *    reference CPU broadcast; global reductions (dot) use a fixed-order tree
 *    and are deterministic run-to-run but differ from BLAS order (tolerance
 *    stated in DESIGN.md).
 *  - operators are NOT re-entrant (same as the reference, which shares
 *    temporaries in its closures): at most one in-flight call per ctx / handle.
 */
#ifndef MXLO_H
#define MXLO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------ */
#define MXLO_OK       0
#define MXLO_EINVAL   1 /* bad argument (null pointer, negative size, bad dtype/flag)      */
#define MXLO_ESHAPE   2 /* shape mismatch -> glue throws LinearOperatorException             */
#define MXLO_EHIP     3 /* a HIP runtime call failed                                         */
#define MXLO_ENOMEM   4 /* device allocation failed                                          */
#define MXLO_ESTATE   5 /* wrong variant for this handle (e.g. damped push on undamped op)   */
#define MXLO_EDOMAIN  6 /* argument outside domain (sigma < 0 -> glue throws ArgumentError)  */
#define MXLO_EREDUCE  7 /* the user all-reduce hook returned non-zero                        */

/* ---- dtypes and flags --------------------------------------------------- */
#define MXLO_F64 0
#define MXLO_F32 1
#define MXLO_C64 2 /* ComplexF64: (re, im) doubles adjacent — Julia's Complex{Float64}; `_c` entry points only */
#define MXLO_C32 3 /* ComplexF32 */

#define MXLO_ALPHA_F64   0x1 /* f32 data: alpha is a Float64 (the alpha-term is evaluated in double) */
#define MXLO_BETA_F64    0x8 /* f32 data: beta  is a Float64 (the beta-term  is evaluated in double) */
#define MXLO_SCALARS_F64 (MXLO_ALPHA_F64 | MXLO_BETA_F64) /* both (see header comment)                */
#define MXLO_D_SCALAR    0x2 /* diag: `d` has ONE element, broadcast (SpectralGradient,
                                src/DiagonalHessianApproximation.jl:226)                      */
#define MXLO_TAIL_BETA   0x4 /* eye: rows [n_min,nrow) receive `beta` (NOT beta*res) when
                                beta != 0 — reference quirk, src/special-operators.jl:42     */
#define MXLO_CONJ_D      0x10 /* complex diag: multiply by conj.(d) — the ctprod! of opDiagonal,
                                 src/special-operators.jl:139-141                             */
#define MXLO_ALPHA_REAL  0x20 /* complex leaves: alpha is a Real (Real*Complex is componentwise in Julia,
                                 not Complex(alpha, 0)*z: signed zeros and non-finite values differ)   */
#define MXLO_BETA_REAL   0x40 /* the same for beta                                            */
#define MXLO_D_REAL      0x80 /* mxlo_hermitian_mul_c: `d` holds n REAL values of the component type
                               * (the reference test passes real.(diag(A)), test_linop.jl:362)  */

/* transposition modes for matrix-carrying leaves */
#define MXLO_OP_N 0 /* prod!   */
#define MXLO_OP_T 1 /* tprod!  */
#define MXLO_OP_C 2 /* ctprod! (== T for real dtypes)                                     */
#define MXLO_OP_J 3 /* conj(M)*v — complex entry points only: what a row-major alias of M' needs */

typedef struct mxlo_ctx mxlo_ctx;     /* opaque: device, stream, reduction workspace, tuning  */
typedef struct mxlo_qn mxlo_qn;       /* opaque: L-BFGS / L-SR1 state resident in HBM         */
typedef struct mxlo_timer mxlo_timer; /* opaque: a hipEvent pair on the ctx stream            */

/* ---- library / context --------------------------------------------------- */
const char *mxlo_version(void);
const char *mxlo_status_string(int32_t status);
const char *mxlo_last_error(void);

/* `stream` is a hipStream_t (or NULL for the device's default stream). The
 * ctx does not own a caller-provided stream. */
int32_t mxlo_ctx_create(int32_t device_id, void *stream, mxlo_ctx **out);
int32_t mxlo_ctx_destroy(mxlo_ctx *ctx);
/* Switching streams inserts an event dependency old -> new (event recorded on the old stream, waited for by the new
 * one), so ALL work issued through one ctx is totally ordered whatever streams it is put on: the reduction workspace,
 * the scalar buffer, the quasi-Newton handles and the exchange slots of the single-launch Householder (one slot set
 * + epoch word per ctx, valid for <= 1 fused launch in flight) are never touched by two kernels at once. Callers that
 * want two applies to OVERLAP use two ctxs (one per stream). tests/test_gpu_leaves.py::
 * test_single_launch_householder_two_streams_one_ctx alternates two streams on one ctx. */
int32_t mxlo_ctx_set_stream(mxlo_ctx *ctx, void *stream);
int32_t mxlo_ctx_sync(mxlo_ctx *ctx);
/* info[0]=device id, [1]=CU count, [2]=workspace bytes, [3]=max reduction columns */
int32_t mxlo_ctx_info(mxlo_ctx *ctx, int64_t info[4]);
/* Launch-geometry / algorithm-variant knobs for sweeps and for tests that compare two device implementations. Keys:
 * "blocks_per_cu", "nt_min_bytes", "red_blocks_per_cu", "graph_direct_max", "house_fused", "house_reverse", "house_inline_n" (two-pass opHouseholder up to this n: the update pass adds up the dots pass's partial sums itself, no finalize launch; 0: never),
 * "cherm_two_pass", "lbfgs_inv_mode", "gemm_tile", "extend_tiles_per_block", "fuse_finalize", "combine_blocks_per_cu",
 * "dots_max_nc", "sp_xcds" (sparse apply: number of L2 domains the chunk order is banded over — 8 = one contiguous part of the chunk table per XCD of an MI355X; default 1 = plain order), "fused_timeout_ms", "fused_debug_drop" (both below), "qn_fused_small" (1: dots + finalize + coefficients of a small quasi-Newton apply in one launch), "qn_fused_max_grid" (most workgroups of that launch, 1..256, default 256: vectors up to 2^19 doubles with 4 vectors per lane; 64 was the round-3 limit), "qn_fused_batch12" (1: with 9 .. 12 panel columns on a short vector that launch takes all columns in one batch — one memory round trip less in each of its two phases), "push_wide", "push_posted" (1: push!'s few doubles of decision state are posted into mapped pinned host memory of the handle by a one-wave kernel and the host polls a sequence word behind them — falling back to a stream synchronisation after 200 us of polling, and to a plain copy from device memory (posting off for that handle from then on) should the posted words not be there after it; 0: hipMemcpyAsync + hipStreamSynchronize; 2: debug — every posting is treated as lost), "qn_persist" (1, default: a quasi-Newton apply at cache-resident sizes — n >= "qn_persist_min_n" (2^19), "qn_persist_min_bytes" (32 MiB) <= panel bytes <= "qn_persist_max_bytes" (448 MiB) — is ONE persistent launch: one 512-thread workgroup per CU owning a contiguous run of the vectors, dots -> grid exchange -> coefficients -> combine; "qn_persist_reverse" (1): its combine phase walks back to front; "qn_persist_prefetch" (0; measured slower at n <= 2^20): x and the first column batch of the first combine chunk are requested before the exchange; "qn_persist_lds" (1, default; round 6): the dots phase parks x of every chunk and the first columns of the combine order in the CU's LDS (18 tiles of 8 KiB) and the combine phase reads them from there — bit-identical to 0; "qn_persist_lds_pad": bytes of unused dynamic LDS per workgroup (a placement experiment, default 0)), "herm_order" (interior strips of opHermitian walked 0: row group by row group, 1: column block by column block — same partial layout, same results), "herm_nt" (-1, default: the strip loads of opHermitian carry the nontemporal hint except for triangles of "herm_dp_min_bytes" (96 MiB) <= bytes < "herm_nt_min_bytes" (384 MiB) — about the size of the Infinity Cache, where the next apply finds part of a default-policy stream still cached; 0 / 1: never / always nontemporal), "house_fused_per_cu" (2, default; round 6: the single-launch Householder may use two co-resident workgroups per CU — vectors up to 2^22 doubles keep their slices of h and v in registers, 24 instead of 40 B/elt; 1: one per CU, up to 2^21), "herm_poll_sleep" (4: the finishers of the single-launch opHermitian look at their last partials every 64 x this many clocks while the strips stream), "herm_strip" (0, default: tiles per strip of opHermitian by size; 1 / 2 / 8 force it — a sweep knob, results do not depend on it beyond rounding of the row partials' grouping), "herm_lds_pad" (0; occupancy experiment: bytes of unused dynamic LDS per workgroup of the opHermitian pass launch, 0..49152), "herm_single" (1, default: opHermitian on full row groups of an aligned matrix whose strict lower triangle has at most "herm_single_max_bytes" (112 MiB: f64 n <= 5120, f32 n <= 7424; "herm_single_max_n" > 0 replaces the rule by n <= that value) is ONE launch — strip workgroups publish their partial sums as self-validating slots, finisher workgroups of the same launch wait for the slots of their rows, add them in the order of the separate finish launch (bit-identical) and re-arm them), "gemv_n_rows" (1, default: dense M*v on an aligned matrix with >= 8 x 16 B of rows per CU runs as ONE launch of row bands — a workgroup owns 16 ... 128 rows across all columns, the column sum never leaves it; 0: the two-launch column-chunk schedule; 8/16/32 x (16 / sizeof(T)): that band height, for sweeps), "kron_fuse" (1, default: both GEMMs of a kron apply whose tiles fit one per CU run in ONE launch — a row block of the first product is produced and consumed by workgroups of one XCD, published through a counter in that XCD's L2, no device-wide barrier; bounded wait + fault word like every single-launch form; 0: two launches; 2: timing experiment without the wait, wrong results), "gemvb_n_rows" (1, default: the block apply M*V of a dense operator runs where M*v takes its 512- / 256-byte row bands as ONE launch of those bands with the block of vectors staged in LDS once per workgroup — no partial workspace, every column bit-identical to M*v on that column; 0: the column-chunk schedule + finish launch), "gemvb_t_lds" (1, default: the transposed block apply of a dense operator with >= 4 columns stages the block of vectors in LDS once per workgroup), "combine_reverse" (0, default: the combine pass of a four-launch apply walks front to back; 1 measured no gain), "push_fused" (1: streaming push! schedules — L-BFGS: new pair held per lane, in-pass inserts; L-SR1: panels once, y - B s never stored, inserts ride in the rebuild; 0: the copies + dots schedules they replaced). Unknown key or out-of-range value -> MXLO_EINVAL.
 * "house_fused" / "qn_fused_small" = 0 is also the setting for MORE than two processes sharing one GPU: the workgroups of a
 * single-launch apply wait for each other, so a launch must be resident as a whole; two of the largest Householder launches
 * (256 workgroups of 179-209 VGPRs: 2^20 < n <= 2^21 doubles; four of those up to 2^20 — since round 6 a launch for 2^21 < n <= 2^22 doubles takes the WHOLE chip itself, two workgroups per CU, "house_fused_per_cu"; and the persistent quasi-Newton apply with LDS parking, "qn_persist_lds", owns every CU's LDS: ONE of either at a time) or two quasi-Newton ones (up to 256
 * workgroups of 166-224 VGPRs since round 4; twelve of the 64-workgroup launches of short vectors) fit on the chip at once, beyond that two launches could each be partly resident and wait for each other —
 * until the bounded wait below ends them. (They are never used with an all-reduce hook installed.)
 * That wait is BOUNDED: "fused_timeout_ms" (default 2000, 1..600000) is how long a workgroup polls for a peer's partial before
 * it gives up, stores NaN and raises the ctx fault word (pinned host memory). The host reads that word, without
 * synchronising, before every single-launch apply and at the end of mxlo_ctx_sync: the call then returns MXLO_EHIP naming the
 * timeout, after draining the stream, re-arming the exchange slots and switching "house_fused" / "qn_fused_small" / "qn_persist" OFF for
 * the ctx (the apply that timed out has stored NaN — repeat it). Before a single-launch apply is issued the library also
 * checks, once per kernel and device, that the occupancy of the kernel lets the whole grid be resident, and falls back to
 * the multi-launch form otherwise. "fused_debug_drop" (default -1) is a TEST HOOK: that workgroup index never publishes
 * its partial, which is how tests/test_gpu_leaves.py provokes the timeout. */
int32_t mxlo_ctx_tune(mxlo_ctx *ctx, const char *key, int64_t value);

/* Row-sharding hook. When set, EVERY global reduction this ctx performs
 * (Householder h'v, L-BFGS/L-SR1 panel dots, push! dots, shifted-solve dots)
 * calls `fn(user, dev_buf, count, stream)` right after the local fixed-order
 * finalize and before any kernel consumes the scalars: `dev_buf` holds `count`
 * doubles in device memory that must be sum-all-reduced in place, stream-
 * ordered on `stream` (e.g. ncclAllReduce(dev_buf, dev_buf, count, ncclDouble,
 * ncclSum, comm, stream)). All ranks then hold bit-identical scalars, which
 * drive the replicated control flow (`ys[k] != 0` skips, push! rejection).
 * The reference has no distributed path; this is the seam for it. */
typedef int32_t (*mxlo_allreduce_fn)(void *user, void *dev_buf, int64_t count, void *stream);
int32_t mxlo_ctx_set_allreduce(mxlo_ctx *ctx, mxlo_allreduce_fn fn, void *user);

/* ---- device-memory helpers for the glue's device-vector type ------------- */
int32_t mxlo_malloc(mxlo_ctx *ctx, int64_t bytes, void **out);
int32_t mxlo_free(mxlo_ctx *ctx, void *p);
int32_t mxlo_memcpy_h2d(mxlo_ctx *ctx, void *dst, const void *src, int64_t bytes);
int32_t mxlo_memcpy_d2h(mxlo_ctx *ctx, void *dst, const void *src, int64_t bytes); /* syncs */
int32_t mxlo_memcpy_d2d(mxlo_ctx *ctx, void *dst, const void *src, int64_t bytes);
int32_t mxlo_memset(mxlo_ctx *ctx, void *p, int32_t byte, int64_t bytes);

/* ---- the allocation / synchronisation contract, observable (test hook) ------------------------------------------
 * The reference's tests assert that a warmed mul! allocates nothing (test/test_lbfgs.jl:180-218). Here the
 * equivalent statement is about the HIP runtime: a warmed mul! / diag! / solve_shifted_system! issues kernel
 * launches and nothing else — no hipMalloc/hipFree, no copy, no stream/device/event synchronisation — and push!
 * issues exactly ONE small device-to-host transfer (its 2-6 doubles) and the one wait that transfer needs. (With the
 * tune key "push_posted" = 1 the transfer is a one-wave kernel writing mapped pinned host memory and the wait is the
 * host polling a sequence word; both are counted below as one D2H copy and one stream synchronisation.)
 * Every allocating / copying / blocking runtime call the library makes is counted (process-wide, monotone):
 *   out[0] hipMalloc+hipHostMalloc   [1] hipFree   [2] H2D copies   [3] D2H copies   [4] D2D copies
 *   out[5] bytes copied D2H          [6] hipStreamSynchronize       [7] hipDeviceSynchronize
 *   out[8] hipEventSynchronize       [9] hipMemsetAsync             [10] kernel launches
 *   out[11] blocking (non-Async) copies
 * Differences of two snapshots around a call give what that call did (tests/test_gpu_contract.py). */
int32_t mxlo_debug_counters(int64_t out[12]);

/* ---- hipGraph capture: launch-bound inner loops --------------------------------------------
 * An apply at small n is 2-4 dependent kernel launches (dots -> finalize -> coefficients -> combine);
 * a Krylov or quasi-Newton inner loop repeats the same sequence on the same buffers thousands of times.
 * Everything the apply entry points enqueue is stream-ordered with no host synchronisation, so a whole
 * sequence of calls can be recorded once and replayed with ONE launch:
 *     mxlo_graph_begin(ctx); mxlo_householder_mul(...); mxlo_qn_mul(...); ...; mxlo_graph_end(ctx, &g);
 *     loop: mxlo_graph_launch(g);
 * Buffers, sizes and alpha/beta are baked in (the data they point to is read at replay time). The ctx
 * stream must not be the default stream (mxlo_ctx_create_stream gives the ctx one it owns). Not
 * capturable: push! and mxlo_memcpy_d2h (host control flow / synchronising) and the first opHermitian
 * apply of a new size (workspace growth) — run those outside, once, before capturing.
 * Staleness: a quasi-Newton apply bakes the handle's slot order, insert position and update count into the
 * recorded launches, and opHermitian bakes the ctx scratch pointer. A graph therefore remembers the generation of
 * every quasi-Newton handle it captured (bumped by push!, reset! and mode changes) and of the ctx scratch
 * (bumped by workspace growth); mxlo_graph_launch returns MXLO_ESTATE — it never replays stale metadata — when
 * any of them changed or a captured handle was destroyed. Re-capture after push!/reset!. */
typedef struct mxlo_graph mxlo_graph;
int32_t mxlo_ctx_create_stream(mxlo_ctx *ctx, void **stream_out); /* non-blocking stream owned by the ctx */
int32_t mxlo_graph_begin(mxlo_ctx *ctx);
int32_t mxlo_graph_end(mxlo_ctx *ctx, mxlo_graph **out);
int32_t mxlo_graph_launch(mxlo_graph *g);   /* on the stream it was captured from; ordered, both ways, with the ctx's
                                             * current stream when mxlo_ctx_set_stream moved the ctx since the capture */
/* info[0] = nodes recorded, info[1] = 1 when the replay re-issues the recorded launches directly (a dependency chain of
 * at most `graph_direct_max` kernel/memset nodes, tune key, default 16: cheaper than hipGraphLaunch on this runtime),
 * 0 when it goes through hipGraphLaunch. Same results either way. */
int32_t mxlo_graph_info(mxlo_graph *g, int64_t info[2]);
int32_t mxlo_graph_destroy(mxlo_graph *g);

/* ---- timing on the ctx stream (bench / roofline evidence) ---------------- */
int32_t mxlo_timer_create(mxlo_ctx *ctx, mxlo_timer **out);
int32_t mxlo_timer_start(mxlo_timer *t);
int32_t mxlo_timer_stop(mxlo_timer *t);
int32_t mxlo_timer_elapsed_ms(mxlo_timer *t, double *ms); /* syncs on the stop event */
int32_t mxlo_timer_destroy(mxlo_timer *t);

/* ======================================================================== */
/*  Leaf closures                                                            */
/* ======================================================================== */

/* mulSquareOpDiagonal! / mulOpDiagonal!  — src/special-operators.jl:125-131,144-151
 * (also the mul! of DiagonalPSB/Andrei/BFGS/SpectralGradient,
 *  src/DiagonalHessianApproximation.jl:37,112,179,226).
 *   i <  n_min : res[i] = (alpha*d[i])*v[i]                    (beta == 0)
 *                res[i] = ((alpha*d[i])*v[i]) + (beta*res[i])  (beta != 0)
 *   n_min <= i < nrow : res[i] = 0   (rectangular form zeroes the tail
 *                                     regardless of beta, :150)
 * Square operator: n_min == nrow. ctprod! on real data is the same call. */
int32_t mxlo_diag_mul(mxlo_ctx *ctx, int32_t dtype, void *res, const void *d, const void *v,
                      int64_t n_min, int64_t nrow, double alpha, double beta, int32_t flags);

/* mulOpEye! — src/special-operators.jl:36-44.
 *   i < n_min: res[i] = alpha*v[i] (+ beta*res[i]);
 *   tail: 0 when beta==0, else `beta` itself (MXLO_TAIL_BETA, reference quirk)
 *   or beta*res[i] (flag clear; used as the generic axpby of prod3!,
 *   src/operations.jl:10-20: `res .= alpha .* Mv .+ beta .* res`). */
int32_t mxlo_eye_mul(mxlo_ctx *ctx, int32_t dtype, void *res, const void *v, int64_t n_min,
                     int64_t nrow, double alpha, double beta, int32_t flags);

/* mulOpZeros! — src/special-operators.jl:102-108: res .= 0  |  res .*= beta. */
int32_t mxlo_zeros_mul(mxlo_ctx *ctx, int32_t dtype, void *res, int64_t nrow, double beta,
                       int32_t flags);

/* mulOpOnes! — src/special-operators.jl:79-85: res .= (alpha*sum(v)) (.+ beta.*res). */
int32_t mxlo_ones_mul(mxlo_ctx *ctx, int32_t dtype, void *res, int64_t nrow, const void *v,
                      int64_t ncol, double alpha, double beta, int32_t flags);

/* fill!(res, value) — `op.d .= one(T)` of reset!(::AbstractDiagonalQuasiNewtonOperator)
 * (src/DiagonalHessianApproximation.jl:71-77) and the glue's Base.fill! on device vectors. */
int32_t mxlo_fill(mxlo_ctx *ctx, int32_t dtype, void *res, int64_t n, double value);

/* `res .*= alpha` of prod3! — src/operations.jl:13-15. */
int32_t mxlo_scale(mxlo_ctx *ctx, int32_t dtype, void *res, int64_t n, double alpha,
                   int32_t flags);

/* ---- complex element types (dtype MXLO_C64 / MXLO_C32) ------------------------------------------------------
 * The reference's own tests of opDiagonal and opHouseholder run on ComplexF64 (test/test_linop.jl:308-318,
 * 511-517). Scalars arrive as (re, im) pairs; MXLO_ALPHA_REAL / MXLO_BETA_REAL mark Real caller scalars,
 * MXLO_ALPHA_F64 / MXLO_BETA_F64 their width next to ComplexF32 data (same promotion rule as the real leaves).
 * Arithmetic is Julia's, component by component, nothing fused:
 *   Complex*Complex = (zr*wr - zi*wi, zr*wi + zi*wr),  Real*Complex = (x*wr, x*wi).
 * Elementwise results are bit-identical to the reference CPU broadcast; the Householder dot is a fixed-order tree. */
/* mulSquareOpDiagonal!/mulOpDiagonal! (src/special-operators.jl:125-131,144-151); MXLO_CONJ_D = ctprod! (:139-141). */
int32_t mxlo_diag_mul_c(mxlo_ctx *ctx, int32_t dtype, void *res, const void *d, const void *v, int64_t n_min,
                        int64_t nrow, double alpha_re, double alpha_im, double beta_re, double beta_im,
                        int32_t flags);
/* mulOpEye! / the generic axpby of prod3! (src/special-operators.jl:36-44, src/operations.jl:18). */
int32_t mxlo_eye_mul_c(mxlo_ctx *ctx, int32_t dtype, void *res, const void *v, int64_t n_min, int64_t nrow,
                       double alpha_re, double alpha_im, double beta_re, double beta_im, int32_t flags);
/* mulOpZeros! (src/special-operators.jl:102-108): res .= 0 | res .*= beta. */
int32_t mxlo_zeros_mul_c(mxlo_ctx *ctx, int32_t dtype, void *res, int64_t nrow, double beta_re, double beta_im,
                         int32_t flags);
/* `res .*= alpha` of prod3! (src/operations.jl:13-15). */
int32_t mxlo_scale_c(mxlo_ctx *ctx, int32_t dtype, void *res, int64_t n, double alpha_re, double alpha_im,
                     int32_t flags);
/* conj!(res) (res == v) and conj.(v) of the wrapper routing (src/adjtrans.jl:127-136,193-204,226-249). */
int32_t mxlo_conj_c(mxlo_ctx *ctx, int32_t dtype, void *res, const void *v, int64_t n);
/* LinearAlgebra.dot(a, b) = sum conj(a_i) b_i into TWO device doubles (re, im); runs the all-reduce hook. */
int32_t mxlo_dot_c(mxlo_ctx *ctx, int32_t dtype, const void *a, const void *b, int64_t n, double *out_dev);
/* Dense LinearOperator(M) on ComplexF64 / ComplexF32 — src/constructors.jl:19-29: prod! = mul!(res, M, v, α, β), tprod!
 * with transpose(M), ctprod! with M' (op_mode MXLO_OP_N / _T / _C; MXLO_OP_J = conj(M)*v). M is m x n column-major
 * with leading dimension ld. Reductions in f64, fixed order. (The reference's "Transpose and adjoint" testset runs on
 * ComplexF64, test/test_linop.jl:385-420.) */
int32_t mxlo_gemv_c(mxlo_ctx *ctx, int32_t dtype, void *res, const void *M, int64_t m, int64_t n, int64_t ld,
                    const void *v, double alpha_re, double alpha_im, double beta_re, double beta_im,
                    int32_t op_mode, int32_t flags);
/* mulHermitian! on complex data — src/linalg.jl:97-103: res = α*(d.*v + L*v + (v'*L)') (+ β*res), L = tril(A,-1) read
 * from the caller's column-major A (only the strict lower triangle is touched); d complex, or real with MXLO_D_REAL
 * (test/test_linop.jl:360-370 builds it from ComplexF64 A and real d). ONE pass over the triangle (row and column
 * partials per strip, fixed-order finish); tune key "cherm_two_pass" = 1 selects the two-pass form (L*v, then L'*v). */
int32_t mxlo_hermitian_mul_c(mxlo_ctx *ctx, int32_t dtype, void *res, const void *d, const void *A, int64_t lda,
                             const void *v, int64_t n, double alpha_re, double alpha_im, double beta_re,
                             double beta_im, int32_t flags);
/* mulHouseholder! (src/linalg.jl:77-83) for complex h: c = 2*dot(h, v) with h conjugated. */
int32_t mxlo_householder_mul_c(mxlo_ctx *ctx, int32_t dtype, void *res, const void *h, const void *v, int64_t n,
                               double alpha_re, double alpha_im, double beta_re, double beta_im, int32_t flags);

/* mulHouseholder! — src/linalg.jl:77-83.
 *   c = 2*dot(h,v);  res[i] = alpha*(v[i] - c*h[i]) (+ beta*res[i]).
 * Two launches + finalize: (A) fixed-order partial dots -> device scalar,
 * [all-reduce hook], (B) update reading the device scalar. No host sync. */
int32_t mxlo_householder_mul(mxlo_ctx *ctx, int32_t dtype, void *res, const void *h,
                             const void *v, int64_t n, double alpha, double beta,
                             int32_t flags);

/* The two phases of mxlo_householder_mul, exposed separately (profiling, and callers that
 * already hold h'v): mxlo_dot is LinearAlgebra.dot(a, b) (src/linalg.jl:79) into ONE device
 * double (fixed-order tree; runs the all-reduce hook); mxlo_householder_apply is the update
 * pass reading that device scalar. */
int32_t mxlo_dot(mxlo_ctx *ctx, int32_t dtype, const void *a, const void *b, int64_t n,
                 double *out_dev);
int32_t mxlo_householder_apply(mxlo_ctx *ctx, int32_t dtype, void *res, const void *h,
                               const void *v, int64_t n, double alpha, double beta,
                               int32_t flags, const double *dot_dev);

/* mulHermitian! — src/linalg.jl:97-103, operator built by opHermitian(d, A)
 * (:109-116) which keeps L = tril(A,-1).
 *   res = alpha*((d.*v + L*v) + L'*v) (+ beta*res)
 * `A` is the ORIGINAL n x n column-major matrix (leading dimension lda); only
 * its strict lower triangle is read, once. */
int32_t mxlo_hermitian_mul(mxlo_ctx *ctx, int32_t dtype, void *res, const void *d,
                           const void *A, int64_t lda, const void *v, int64_t n, double alpha,
                           double beta, int32_t flags);
/* The same operator applied to the k columns of a matrix: res[:, c] = α((d .* V[:, c] + L V[:, c]) + L' V[:, c]) + β res[:, c].
 * An EXTENSION of the reference, not a match: `mul!(res::Matrix, opHermitian(d, A), V::Matrix, α, β)` hands the matrices to
 * the closure (src/operations.jl:34-36), whose `(...)[:]` (src/linalg.jl:99-101) flattens the n x k product, so upstream the
 * call throws DimensionMismatch for k > 1. Offered because block Krylov callers need it; the host sides apply complex data
 * column by column through mxlo_hermitian_mul_c (no block entry point for complex element types).
 * The strict lower triangle is read ONCE per chunk of up to 4 columns instead of once per column (block Krylov shapes);
 * per column the arithmetic and the order of every addition are those of mxlo_hermitian_mul, so the block apply is
 * BIT-IDENTICAL to k single applies. res, V: column-major with leading dimensions ldr, ldv >= n. Float64 / Float32. */
int32_t mxlo_hermitian_mul_block(mxlo_ctx *ctx, int32_t dtype, void *res, int64_t ldr, const void *d, const void *A, int64_t lda,
                                 const void *V, int64_t ldv, int64_t n, int64_t k, double alpha, double beta, int32_t flags);

/* mulRestrict! — src/special-operators.jl:167-169: res .= view(v, I); alpha and
 * beta are IGNORED by the reference and are therefore not parameters.
 * Pure data movement: bit-exact, `elem_size` in {4, 8, 16} bytes.
 *  idx form   : I is a device Int64 vector of 1-BASED indices as Julia stores them.
 *  range form : I = start:step:start+step*(len-1) (UnitRange / StepRange), 1-based. */
int32_t mxlo_gather(mxlo_ctx *ctx, int32_t elem_size, void *res, const void *v, int64_t nv,
                    const int64_t *idx, int64_t nidx);
int32_t mxlo_gather_range(mxlo_ctx *ctx, int32_t elem_size, void *res, const void *v,
                          int64_t nv, int64_t start, int64_t step, int64_t len);

/* multRestrict! — src/special-operators.jl:171-174: res .= 0; res[I] = u.
 * Duplicate indices: the reference's sequential loop makes the LAST write win.
 * The glue resolves that at operator construction (indices are construction-
 * time data): it passes the deduplicated index list plus `pos` (0-based
 * position in `u` of the surviving write for each kept index; NULL when there
 * are no duplicates, meaning pos[k] == k). */
int32_t mxlo_scatter_zero(mxlo_ctx *ctx, int32_t elem_size, void *res, int64_t nres,
                          const void *u, const int64_t *idx, const int64_t *pos, int64_t nidx);
/* The same statement pair for a plan whose indices are STRICTLY INCREASING (the glue sorts its last-write-wins plan
 * once; pos[k], 0-based or NULL for the identity, is the position in u of sorted entry k): segment-owner kernel —
 * each workgroup builds one output tile (zeros and values) in LDS and writes it once with 16-byte stores, so res
 * is written exactly once and u is the only scattered access. Bit-exact like mxlo_scatter_zero. An index list that
 * is not strictly increasing gives an unspecified (memory-safe) result: the CALLER guarantees the order. */
int32_t mxlo_scatter_zero_sorted(mxlo_ctx *ctx, int32_t elem_size, void *res, int64_t nres,
                                 const void *u, const int64_t *idx, const int64_t *pos, int64_t nidx);
int32_t mxlo_scatter_zero_range(mxlo_ctx *ctx, int32_t elem_size, void *res, int64_t nres,
                                const void *u, int64_t start, int64_t step, int64_t len);

/* Index plans (round 5): a STRICTLY INCREASING index set I ⊂ 1..n kept as one bit per element of the long vector plus
 * the number of selected elements before every 64-bit mask word (k-th index <-> k-th set bit). Neither
 * mulRestrict! (res = v[I], src/special-operators.jl:167-169) nor multRestrict! (res .= 0; res[I] = u, :171-174) then
 * reads the index list: 1/4 byte per element of the LONG vector replaces 8 bytes per index, and both applies are
 * streaming passes over the long vector (extension: res written exactly once with 16-byte stores, no search, no LDS
 * tile; restriction: 16-byte groups of v read only where a bit is set, the selected elements stored at consecutive
 * addresses). Worth it when nidx >= n / 32; bit-exact like mxlo_gather / mxlo_scatter_zero (pure data movement).
 * The plan is built ONCE, at operator construction, from HOST indices (MXLO_EDOMAIN when they are not strictly
 * increasing within 1..n) and is bound to its ctx's device. `pos` of mxlo_scatter_zero_plan: as in
 * mxlo_scatter_zero_sorted (device array, 0-based position in u of the k-th sorted index; NULL = identity). */
typedef struct mxlo_index_plan mxlo_index_plan;
int32_t mxlo_index_plan_create(mxlo_ctx *ctx, const int64_t *idx_host, int64_t nidx, int64_t n, mxlo_index_plan **out);
int32_t mxlo_index_plan_destroy(mxlo_index_plan *plan);
int32_t mxlo_gather_plan(mxlo_ctx *ctx, int32_t elem_size, void *res, const void *v, int64_t nv, const mxlo_index_plan *plan);
int32_t mxlo_scatter_zero_plan(mxlo_ctx *ctx, int32_t elem_size, void *res, int64_t nres, const void *u, const int64_t *pos,
                               const mxlo_index_plan *plan);

/* Row-shard staging for collectives that move vectors (row-sharded dense LinearOperator / opHermitian, SURVEY §8f-4).
 * Shard r of an n-vector owns rows [lo(r), lo(r)+len(r)), lo(r) = r*q + min(r, rem), len(r) = q + (r < rem),
 * q = n / world, rem = n % world; the wire format of all-gather / reduce-scatter is `world` slots of
 * pad = ceil(n / world) elements. One launch either way:
 *   MXLO_SHARD_PACK   padded[r*pad + k] = full[lo(r)+k] for k < len(r) and lo(r)+k < nvalid, else 0
 *                     (rows >= nvalid count as zeros and are not read; nvalid < 0 means n)
 *   MXLO_SHARD_UNPACK full[lo(r)+k] = padded[r*pad + k] for k < len(r)
 * elem_size 4, 8 or 16 bytes (bit copies, like gather/scatter). */
#define MXLO_SHARD_PACK 0
#define MXLO_SHARD_UNPACK 1
int32_t mxlo_shard_stage(mxlo_ctx *ctx, int32_t elem_size, void *dst, const void *src, int64_t n,
                         int32_t world, int64_t nvalid, int32_t direction);

/* BlockDiagonalOperator prod!/tprod!/ctprod! — src/special-operators.jl:258-289.
 * The reference loops over blocks issuing one inner mul! per block (1024
 * launches at BASELINE config 4); here the whole operator is ONE launch over a
 * device-resident descriptor table built once at construction. */
#define MXLO_BLK_DIAG  0 /* opDiagonal block: data = d (m == n)                            */
#define MXLO_BLK_DENSE 1 /* plain matrix block m x n, column-major, leading dimension ld   */
#define MXLO_BLK_EYE   2 /* opEye(n)                                                       */
#define MXLO_BLK_ZEROS 3 /* opZeros(m,n)                                                   */
#define MXLO_BLK_CSC   4 /* sparse block m x n: data = the block's mxlo_csc handle (below), which must
                            outlive the block-diagonal operator; element type = the operator's      */
typedef struct mxlo_block_desc {
  int32_t kind;     /* MXLO_BLK_*                                   */
  int32_t reserved;
  int64_t row_off;  /* first row of this block in res (0-based)     */
  int64_t col_off;  /* first column of this block in v (0-based)    */
  int64_t m, n;     /* block size                                   */
  const void *data; /* device pointer (d or matrix), NULL for eye/zeros */
  int64_t ld;       /* leading dimension (dense)                    */
} mxlo_block_desc;
typedef struct mxlo_blockdiag mxlo_blockdiag; /* opaque: device descriptor + tile table */
int32_t mxlo_blockdiag_create(mxlo_ctx *ctx, int32_t dtype, const mxlo_block_desc *blocks,
                              int64_t nblocks, mxlo_blockdiag **out);
int32_t mxlo_blockdiag_mul(mxlo_blockdiag *bd, void *res, const void *v, double alpha,
                           double beta, int32_t op_mode, int32_t flags);
int32_t mxlo_blockdiag_destroy(mxlo_blockdiag *bd);

/* kron(A,B) prod!/tprod!/ctprod! — src/kron.jl:14-40.
 *   N: X = reshape(x, q, n);  res = alpha*vec(B*X*transpose(A)) (+ beta*res)
 *   T/C: X = reshape(x, p, m); res = alpha*vec(transpose(B)*X*A) (+ beta*res)
 * A is m x n (lda), B is p x q (ldb), both dense column-major on the device.
 * Two f64/f32 GEMMs on the matrix cores (v_mfma_f64_16x16x4_f64 /
 * v_mfma_f32_16x16x4_f32) with the alpha/beta epilogue fused in the second; in T/C mode A and B are read
 * transposed IN PLACE (K-contiguous operand layout of the GEMM kernel) — no transposed copies.
 * `work` must hold max(p*n, q*m) elements (the glue allocates it once at
 * construction, like the reference's compose temporaries, src/operations.jl:149-151). */
int32_t mxlo_kron_mul(mxlo_ctx *ctx, int32_t dtype, void *res, const void *A, int64_t m,
                      int64_t n, int64_t lda, const void *B, int64_t p, int64_t q, int64_t ldb,
                      const void *x, void *work, double alpha, double beta, int32_t op_mode,
                      int32_t flags);

/* The same with a transposition flag PER FACTOR: kron(opA(A), opB(B)) where opX is the stored column-major matrix
 * (trans == 0) or its transpose (trans != 0). A is stored am x an (lda), B is stored bp x bq (ldb).
 *   (0,0) = prod!, (1,1) = tprod!/ctprod! of src/kron.jl:14-40; the mixed cases let the glue hand over a row-major
 *   (= transposed-in-place) factor without copying it. `work` must hold rows(opA) * cols(opB) elements.
 * Both GEMMs run the DMA kernel whenever pointers are 16-byte aligned and the leading dimensions / contiguous
 * extents are multiples of 16 bytes; anything else falls back to a generic tile kernel. */
int32_t mxlo_kron_mul_ex(mxlo_ctx *ctx, int32_t dtype, void *res, const void *A, int64_t am, int64_t an,
                         int64_t lda, int32_t trans_a, const void *B, int64_t bp, int64_t bq, int64_t ldb,
                         int32_t trans_b, const void *x, void *work, double alpha, double beta, int32_t flags);

/* kron(A, B) on ComplexF64 / ComplexF32 data — src/kron.jl:10-40 with complex factors (test/test_kron.jl:3-8 pairs a
 * Float64 A with a ComplexF64 B). The factors arrive as REAL PLANES (the glue splits a complex factor once, a real
 * factor is passed as it is with a NULL imaginary plane): each complex product is 4 (2 for a real factor) real GEMMs
 * on the MFMA kernel of mxlo_kron_mul_ex. mode_a / mode_b: bit 0 = take the stored planes transposed, bit 1 =
 * conjugate (tprod! = 1, ctprod! = 3, a row-major alias flips bit 0). x, res: complex vectors; work: real scalars,
 * 2*(q*n + m*q + p*m) + 24 of them with (m x n) = opA, (p x q) = opB (each plane is placed 16-byte aligned). */
int32_t mxlo_kron_mul_c(mxlo_ctx *ctx, int32_t dtype, void *res, const void *Ar, const void *Ai, int64_t am, int64_t an,
                        int64_t lda, int32_t mode_a, const void *Br, const void *Bi, int64_t bp, int64_t bq, int64_t ldb,
                        int32_t mode_b, const void *x, void *work, double alpha_re, double alpha_im, double beta_re,
                        double beta_im, int32_t flags);
/* The same product in Gauss / Karatsuba form — THREE real GEMMs per complex product instead of four:
 *   (a + ib)(c + id):  k1 = (a + b) c,  k2 = a (d - c),  k3 = b (c + d);  re = k1 - k3,  im = k1 + k2.
 * Same arguments and semantics as mxlo_kron_mul_c (a real factor still costs its two plain GEMMs); results agree with
 * it to rounding (normwise: the three-multiplication form is not componentwise stable, which the reference's
 * 1e-12 * ||K||_1 criterion, test/test_kron.jl:35, does not ask for). As / Bs: the factor-sum planes a + s*b of A / B
 * (s = -1 when bit 1 of the factor's mode conjugates it, else +1; CONTIGUOUS, leading dimension = rows; form them with
 * mxlo_plane_sum) as the glue caches them per factor state and sign — or NULL: formed per call in `work` (the library
 * keeps no factor state). `work` must be 16-byte aligned and hold mxlo_kron_c3_work_size(am, an, mode_a, bp, bq, mode_b)
 * real scalars: the four planes of x (re, im, im - re, re + im), the planes of the intermediate U (also im - re, re + im),
 * the three GEMM outputs of a stage and one factor-sum plane. The last stage's re / im are formed inside the join pass. */
int64_t mxlo_kron_c3_work_size(int64_t am, int64_t an, int32_t mode_a, int64_t bp, int64_t bq, int32_t mode_b);
int32_t mxlo_plane_sum(mxlo_ctx *ctx, int32_t dtype, void *out, const void *a, const void *b, int64_t rows, int64_t cols,
                       int64_t ld, double sign);   /* out = a + sign*b (dtype: the COMPLEX type whose planes these are) */
int32_t mxlo_kron_mul_c3(mxlo_ctx *ctx, int32_t dtype, void *res, const void *Ar, const void *Ai, const void *As, int64_t am,
                         int64_t an, int64_t lda, int32_t mode_a, const void *Br, const void *Bi, const void *Bs, int64_t bp,
                         int64_t bq, int64_t ldb, int32_t mode_b, const void *x, void *work, double alpha_re, double alpha_im,
                         double beta_re, double beta_im, int32_t flags);
/* A REAL operator applied to complex vectors (eltype(op) = Float64, x::Vector{ComplexF64}: test/test_kron.jl
 * "issue110"; Julia runs the generic closure on the complex vectors). The glue applies the real operator to the two
 * planes: mxlo_split_c writes re[i], im[i] of x; mxlo_join_c computes res = α*(re + i*im) (+ β*res) with complex or
 * Real α, β (flags as for the other _c entry points). dtype names the COMPLEX type; re / im are its component type;
 * im == NULL stands for zeros (a real vector handed to a complex operator: `aopA * rand(5)`, test/test_adjtrans.jl:31-34). */
int32_t mxlo_split_c(mxlo_ctx *ctx, int32_t dtype, void *re, void *im, const void *x, int64_t n);
int32_t mxlo_join_c(mxlo_ctx *ctx, int32_t dtype, void *res, const void *re, const void *im, int64_t n,
                    double alpha_re, double alpha_im, double beta_re, double beta_im, int32_t flags);
/* kron(A, B) when BOTH factors are diagonal operators (opDiagonal / opEye; pass NULL for an identity
 * factor): the fused row/col index-decomposition form of src/kron.jl:14-22,
 *   res[r + c*p] = alpha*(dB[r]*(x[r + c*p]*dA[c])) (+ beta*res[r + c*p]),  A is m x m, B is p x p.
 * Symmetric, so tprod!/ctprod! are the same call. One HBM pass, no GEMM. */
int32_t mxlo_kron_diag_mul(mxlo_ctx *ctx, int32_t dtype, void *res, const void *dA, int64_t m,
                           const void *dB, int64_t p, const void *x, double alpha, double beta,
                           int32_t flags);

/* Dense LinearOperator(M) prod!/tprod! — src/constructors.jl:19-29 delegates to
 * LinearAlgebra.mul!; needed for dense blocks and for kron of operators. GEMV. */
int32_t mxlo_gemv(mxlo_ctx *ctx, int32_t dtype, void *res, const void *M, int64_t m, int64_t n,
                  int64_t ld, const void *v, double alpha, double beta, int32_t op_mode,
                  int32_t flags);

/* mul!(res::Matrix, LinearOperator(M), V::Matrix, α, β) — src/operations.jl:34-36 hands the matrices to the closure of
 * src/constructors.jl:19-29 (a GEMM in the reference; test/test_linop.jl:64-76). res (nres x k, leading dimension ldr) =
 * alpha * op(M) * V (nin x k, ldv) + beta * res with op_mode MXLO_OP_N / _T (_C == _T for real data); M is m x n
 * column-major (ld). Tall-skinny blocks: M is read ONCE per chunk of up to 8 columns of V (HBM-bound on M), f64
 * accumulation, fixed-order sums. */
int32_t mxlo_gemv_block(mxlo_ctx *ctx, int32_t dtype, void *res, int64_t ldr, const void *M, int64_t m, int64_t n,
                        int64_t ld, const void *V, int64_t ldv, int64_t k, double alpha, double beta, int32_t op_mode,
                        int32_t flags);

/* Sparse LinearOperator(M::SparseMatrixCSC) prod!/tprod!/ctprod! — src/constructors.jl:19-29 hands M to
 * LinearAlgebra.mul!(res, M, v, α, β), which for a SparseMatrixCSC is the SparseArrays stdlib (Julia >= 1.10,
 * stdlib/SparseArrays/src/linalg.jl: `_spmatmul!` — res scaled by β, or zero-filled when β == 0, then one column sweep
 * `res[rowval[k]] += nzval[k] * (v[col] * α)`; `_At_or_Ac_mul_B!` — per column `tmp += nzval[k]' * v[rowval[k]]`, then
 * `res[col] += tmp * α`). The reference builds such operators in test/test_linop.jl:743-756 (a sprand block of
 * BlockDiagonalOperator), test/test_kron.jl:3-8 (sparse kron factors) and test/test_cat.jl.
 * mxlo_csc_create takes the three arrays of a SparseMatrixCSC{T,Int64} AS STORED (device memory; index_base 1 for
 * Julia, 0 for torch's sparse_csc tensors): colptr[n+1], rowval[nnz] (any order within a column, duplicates summed like
 * the reference's loops do), nzval[nnz]. It validates them (monotone colptr, indices in range — what the reference's
 * constructor checks), keeps 32-bit 0-based copies of the structure and builds the compressed-ROW view that A*x needs
 * (a stable counting sort on the host: columns of a row stay ascending). m, n and nnz must be below 2^31.
 * Values: Aᵀ*x / A'*x read `nzval` IN PLACE on every apply; A*x reads a row-ordered SNAPSHOT of the values taken at
 * create — call mxlo_csc_refresh(h) (one gather pass, stream-ordered) after changing nzval in place. The sparsity
 * pattern is fixed for the life of the handle; the handle keeps `nzval` (not colptr / rowval) referenced.
 * mxlo_csc_mul: res = alpha * op(A) * v + beta * res, op_mode MXLO_OP_N / _T / _C (_C == _T: real element types);
 * beta == 0 never reads res. Both modes are a row gather-reduce on a compressed-row view cut, at create, into chunks of
 * <= 2048 stored entries (whole rows; a row above 512 entries has chunks of its own): a workgroup streams its chunk
 * (coalesced value / index loads) into LDS and walks the rows from there in a fixed order, adjacent lanes gathering x for
 * adjacent rows —
 * HBM traffic and balance do not depend on the row-length distribution, results are run-to-run bit-identical; against the
 * reference the difference is rounding order only (tests: 1e-13 |A||v| Float64, 2e-6 Float32). A row (column, for the
 * transposed modes) with more than 2048 entries is summed piecewise and a second, tiny launch adds its pieces in order.
 * mxlo_csc_info: info = {m, n, nnz, chunks of A*x, chunks of Aᵀ*x, rows above 2048 entries, columns above 2048
 * entries, 2048}. A handle with such rows / columns cannot be a block of the ONE-launch block-diagonal operator. */
typedef struct mxlo_csc mxlo_csc;
int32_t mxlo_csc_create(mxlo_ctx *ctx, int32_t dtype, int64_t m, int64_t n, const int64_t *colptr,
                        const int64_t *rowval, const void *nzval, int32_t index_base, mxlo_csc **out);
int32_t mxlo_csc_refresh(mxlo_csc *h);
int32_t mxlo_csc_mul(mxlo_csc *h, void *res, const void *v, double alpha, double beta, int32_t op_mode,
                     int32_t flags);
/* Complex element types (dtype MXLO_C64 / MXLO_C32 at create; test/test_linop.jl:44 and test/test_cat.jl:5-25 build operators
 * from `simple_sparse_matrix(ComplexF64, …)`): the same chunked sweep on 16- / 8-byte elements, component-wise complex
 * products accumulated in two doubles. op_mode MXLO_OP_N = A*v, _T = transpose(A)*v (values as stored), _C = A'*v (values
 * conjugated in the sweep). Scalars as (re, im) pairs with MXLO_ALPHA_REAL / MXLO_BETA_REAL and the width flags, like every
 * `_c` entry point. mxlo_csc_mul / mxlo_csc_mul_block refuse a complex handle and mxlo_csc_mul_c a real one (MXLO_EINVAL);
 * a complex handle cannot be a block of the (real) one-launch block-diagonal operator. */
int32_t mxlo_csc_mul_c(mxlo_csc *h, void *res, const void *v, double alpha_re, double alpha_im, double beta_re, double beta_im,
                       int32_t op_mode, int32_t flags);
/* mul!(res::Matrix, LinearOperator(A), V::Matrix, α, β) (src/operations.jl:34-36: the closure is handed the matrices —
 * a sparse-times-dense product in the reference): res (nres x k, column-major, leading dimension ldr) = alpha * op(A) * V
 * (nin x k, ldv) + beta * res. A's chunks are streamed into LDS once per group of up to 8 columns and walked once per
 * column: A is read once per 8 columns, each column's result equals the single-vector apply bit for bit. */
int32_t mxlo_csc_mul_block(mxlo_csc *h, void *res, int64_t ldr, const void *V, int64_t ldv, int64_t k, double alpha,
                           double beta, int32_t op_mode, int32_t flags);
int32_t mxlo_csc_info(mxlo_csc *h, int64_t info[8]);
/* Host-only (no device): the chunk table the library builds for a 0-based row-pointer array ptr_host[nrows + 1], as rows
 * {first entry, entries, first row, rows, kind (0 whole rows, 1 one long row, 2 a piece of a row beyond a chunk), carry
 * slot} in out[cap][6] (NULL: counts only). tests/test_abi_and_host.py checks the invariants of the decomposition with it. */
int32_t mxlo_debug_csc_chunks(const int64_t *ptr_host, int64_t nrows, int64_t *out, int64_t cap, int64_t *nchunks,
                              int64_t *nlong, int64_t *ncarry);
int32_t mxlo_csc_destroy(mxlo_csc *h);

/* push!(B, s, y) of the diagonal quasi-Newton operators — src/DiagonalHessianApproximation.jl:
 * DiagonalPSB :45-64, DiagonalAndrei :117-139, DiagonalBFGS :236-249, SpectralGradient :190-199
 * (their mul! is mxlo_diag_mul, :37,112,179,226; SpectralGradient keeps ONE device element in `d` and
 * multiplies with MXLO_D_SCALAR). One fused reduction pass over s, y (, d), the scalar recurrence on
 * the host in the reference's statement order and eltype, one update pass over d.
 * *status = 1 when s == 0: the reference throws ErrorException("Cannot update DiagonalQN operator
 * with s=0") and leaves d untouched; so does this. Costs one 48-byte D2H (the error is host control flow). */
#define MXLO_DQN_PSB      0
#define MXLO_DQN_ANDREI   1
#define MXLO_DQN_BFGS     2
#define MXLO_DQN_SPECTRAL 3
int32_t mxlo_diagqn_push(mxlo_ctx *ctx, int32_t dtype, int32_t kind, void *d, const void *s,
                         const void *y, int64_t n, int32_t *status);

/* ======================================================================== */
/*  Quasi-Newton operators (state resident in HBM)                           */
/* ======================================================================== */
#define MXLO_QN_LBFGS_INV 0 /* InverseLBFGSOperator — src/lbfgs.jl:112-158 */
#define MXLO_QN_LBFGS_FWD 1 /* LBFGSOperator        — src/lbfgs.jl:168-206 */
#define MXLO_QN_LSR1      2 /* LSR1Operator         — src/lsr1.jl:86-111   */

/* inverse two-loop evaluation strategy (ctx tune key "lbfgs_inv_mode" is the default) */
#define MXLO_INV_TWOPASS 0 /* panel form: one dots pass, m x m recurrences on Gram
                              matrices kept up to date by push!, one combine pass      */
#define MXLO_INV_REFORDER 1 /* reference statement order: 2m chained fused axpy+dot      */

/* LBFGSData / LSR1Data constructors — src/lbfgs.jl:26-57, src/lsr1.jl:19-34.
 * mem is clamped to >= 1 like the reference. Two state layouts, chosen by mem, with identical semantics:
 *   mem <= 64 (inverse) / <= 32 (forward, L-SR1): coefficient recurrences run in LDS of ONE workgroup and the
 *             slot metadata travel as kernel arguments (the launch-bound regime the reference's defaults live in);
 *   larger mem, up to 4096: Gram matrices, recurrences and slot order live in HBM (block-wide coefficient
 *             kernels, O(mem^2) doubles); the forward operator always uses the compact representation and
 *             MXLO_PUSH_REFORDER is refused with MXLO_ESTATE there.
 * mem > 4096 is MXLO_EINVAL with a message naming the limit (n x mem panels are the caller's memory budget:
 * 4 panels x n x mem x sizeof(T)). Panels s,y(,a,b) are n x mem
 * column-major allocations owned by the handle; shifted_p is NOT allocated
 * eagerly (the reference allocates n x 2mem at :53; the coefficient-space
 * solve does not need it). */
int32_t mxlo_qn_create(mxlo_ctx *ctx, int32_t kind, int32_t dtype, int64_t n, int64_t mem,
                       int32_t scaling, int32_t damped, double sigma2, double sigma3,
                       mxlo_qn **out);
int32_t mxlo_qn_destroy(mxlo_qn *h);

/* push!(op, s, y) — src/lbfgs.jl:269-287 -> push_common! :210-255;
 *                   src/lsr1.jl:119-184.
 * `accepted` receives 1/0 (the reference silently returns `op` on rejection).
 * Needs the host to see ys (rejection is host control flow): one 2-6 double D2H.
 * s and y may live anywhere, including inside this handle's own panels (a column from mxlo_qn_column, even the slot about
 * to be replaced): the streaming schedules ("push_fused"), which read the pair while the slot is being written, are
 * taken only when [s, s+n) and [y, y+n) overlap none of the handle's allocations — otherwise the pair is copied into
 * its slots first, as the reference does (s[insert] .= s).
 * L-SR1: the accept / reject tests of src/lsr1.jl:131-141 compare r's, y's, |y - s/sf| with thresholds of the order eps,
 * while r's = y's - s'Bs carries an absolute rounding error of the order eps (|y| + |B s|) |s|. The streaming schedule
 * (a_k's from Gram data) and the apply-based one (dots on the stored a_k) round differently, so a push that does not
 * clear its thresholds by 2^8 such error budgets is re-evaluated by the apply-based schedule, whose decision stands. */
int32_t mxlo_qn_push(mxlo_qn *h, const void *s, const void *y, int32_t *accepted);
/* push!(op, s, y, Bs) forward damped — src/lbfgs.jl:289-323. `Bs` is caller scratch (n). */
int32_t mxlo_qn_push_damped_fwd(mxlo_qn *h, const void *s, const void *y, void *Bs,
                                int32_t *accepted);
/* push!(op, s, y, alpha, g, Bs) inverse damped — src/lbfgs.jl:325-357. NOTE: like the
 * reference (`y .= ...`, :351) this OVERWRITES the caller's y when damping triggers. */
int32_t mxlo_qn_push_damped_inv(mxlo_qn *h, const void *s, void *y, double alpha,
                                const void *g, void *Bs, int32_t *accepted);

/* lbfgs_multiply / lsr1_multiply — src/lbfgs.jl:117-154, 173-202; src/lsr1.jl:89-107. */
int32_t mxlo_qn_mul(mxlo_qn *h, void *res, const void *x, double alpha, double beta,
                    int32_t flags);

/* ShiftedOperator(H, sigma) with H a quasi-Newton operator — shifted_prod! src/shifted_operators.jl:16-25:
 *   mul!(res, H, x, alpha, beta);  iszero(sigma) || iszero(alpha) || axpy!(alpha*sigma, x, res)
 * fused into the combine pass of the apply (x is already in registers there): one launch and 3 vector
 * passes fewer, with the SAME per-element rounding sequence as the two separate calls (bit-identical to
 * mxlo_qn_mul followed by the axpy in T arithmetic). H is symmetric: tprod!/ctprod! are the same call. */
int32_t mxlo_qn_mul_shifted(mxlo_qn *h, void *res, const void *x, double alpha, double beta,
                            double sigma, int32_t flags);

/* solve_shifted_system!(x, B, b, sigma) — src/utilities.jl:207-248 (forward L-BFGS only);
 * ldiv! (:281-289) is sigma = 0. sigma < 0 -> MXLO_EDOMAIN (reference: ArgumentError). */
int32_t mxlo_qn_solve_shifted(mxlo_qn *h, void *x, const void *b, double sigma);

/* diag!(op, d) — src/lbfgs.jl:379-395 (forward only), src/lsr1.jl:196-211. */
int32_t mxlo_qn_diag(mxlo_qn *h, void *d);

/* reset!(data) — src/lbfgs.jl:401-415, src/lsr1.jl:217-228 (counters live in the glue). */
int32_t mxlo_qn_reset(mxlo_qn *h);

/* Fields the reference's tests read (test_lbfgs.jl:16,45-46,64-65,70):
 *   scalars[0]=insert (1-based), [1]=scaling_factor, [2]=opnorm_upper_bound,
 *   [3]=mem, [4]=n; per-slot arrays of length mem (may be NULL):
 *   ys, aux = alpha (inverse) | norm_b (forward) | as (L-SR1). */
int32_t mxlo_qn_get_scalars(mxlo_qn *h, double scalars[5], double *ys, double *aux);
/* Device pointer of one panel column, k 0-based: which = 0:s 1:y 2:a 3:b. */
int32_t mxlo_qn_column(mxlo_qn *h, int32_t which, int64_t k, void **out);
/* Evaluation strategy for the inverse two-loop (MXLO_INV_*). */
int32_t mxlo_qn_set_mode(mxlo_qn *h, int32_t mode);
/* Forward L-BFGS push!: how the a_k panel is rebuilt (src/lbfgs.jl:236-250).
 *   MXLO_PUSH_GRAM     (default) coefficient-space recurrence on the Gram matrices S'S, Y'S kept up
 *                      to date by push! (3m dots), then ONE pass A = [S B]*C over the panels;
 *   MXLO_PUSH_REFORDER the reference's statement order: O(m^2) dot/axpy passes over n;
 *   MXLO_PUSH_COMPACT  (forward L-BFGS) the same Gram recurrence, but the a_k panel is NOT formed: a_k = [S B]*c_k
 *                      stays implicit and mul! evaluates x/gamma + [S B]*w with w = -C'(C d) (+ d), d = [S B]'x —
 *                      the same two panel passes per apply, and push! costs only its 3m dots. diag!,
 *                      solve_shifted_system! and mxlo_qn_column materialise the panel on demand (one pass). */
#define MXLO_PUSH_GRAM 0
#define MXLO_PUSH_REFORDER 1
#define MXLO_PUSH_COMPACT 2
int32_t mxlo_qn_set_push_mode(mxlo_qn *h, int32_t mode);

#ifdef __cplusplus
}
#endif
#endif /* MXLO_H */
