// gemm_glds.h — the MFMA GEMM behind kron(A,B) (src/kron.jl:14-40): C = alpha * opA(A) * B'^T (+ beta*C).
//
// One kernel template, gfx950 only:
//  * operands go global -> LDS with `global_load_lds_dwordx4` (LDS-DMA: no staging VGPRs, no ds_write pass),
//    into a ring of NST stages; a stage is issued NST-1 K-slabs ahead and waited for with a COUNTED
//    `s_waitcnt vmcnt(N)` + one raw `s_barrier` per slab, so later slabs stay in flight across the barrier;
//  * the LDS image is lane-linear per DMA instruction (hardware rule), so bank conflicts are avoided by an
//    XOR swizzle applied to the per-lane SOURCE address and to the fragment reads alike — no padding; the swizzle
//    is chosen per operand for the ds_read width that fetches its fragments (see SWA / SWB);
//  * v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32, a wave owns an (TM/WM) x (TN/WN) sub-tile;
//  * the A operand may be M-contiguous (A stored M x K, prod!) or K-contiguous (A stored K x M: tprod!/ctprod!
//    of kron read A and B transposed in place, src/kron.jl:24-40); B' is always N-contiguous (stored N x K);
//  * M, N edges: out-of-range rows/columns read clamped (valid, unused) addresses and are not stored;
//    the K tail runs fewer 4-deep MFMA steps and zeroes the <= 3 slack k-rows in LDS after the DMA landed;
//  * XCD-aware tile order: the 8 XCDs (workgroup id mod 8) each walk a compact band of tiles so the A / B slabs
//    they share stay in that XCD's L2.
#pragma once
#include <utility>

#include "common.h"

namespace mxlo {

typedef double gl_f64x4 __attribute__((ext_vector_type(4)));
typedef float gl_f32x4 __attribute__((ext_vector_type(4)));

template <typename T>
struct GlMfma;
template <>
struct GlMfma<double> {
  using Acc = gl_f64x4;
  static __device__ __forceinline__ Acc run(double a, double b, Acc c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  // C/D layout: col = lane & 15, row = (lane >> 4) + 4*reg
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <>
struct GlMfma<float> {
  using Acc = gl_f32x4;
  static __device__ __forceinline__ Acc run(float a, float b, Acc c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  // C/D layout: col = lane & 15, row = 4*(lane >> 4) + reg
  static __device__ __forceinline__ int row(int lane, int r) { return 4 * (lane >> 4) + r; }
};

template <int N>
__device__ __forceinline__ void gl_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct GlShape {
  int M, N, K;
  int gx, gy;   // tiles along M, N
};

// Tile order: workgroup `id` runs on XCD id % 8. Each XCD gets a contiguous range of the band-major tile order
// (bands of SH tile rows), i.e. a compact sub-grid. Bijective for any tile count.
__device__ __forceinline__ void gl_tile_of(int id, int gx, int gy, int &tx, int &ty) {
  constexpr int SH = 4;
  const int nt = gx * gy, q = nt >> 3, r = nt & 7;
  const int xcd = id & 7, local = id >> 3;
  const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  const int per_band = SH * gy;
  const int band = t / per_band, rem = t - band * per_band;
  const int bh = (gx - band * SH) < SH ? (gx - band * SH) : SH;
  tx = band * SH + rem % bh;
  ty = rem / bh;
}

// AK: the A operand is K-contiguous (element (i,k) at A[k + i*lda]); otherwise M-contiguous (A[i + k*lda]).
// PAIR: a wave's 16-row (16-column) MFMA tiles are interleaved in pairs — tile 2g covers rows wm + 32g + 2*l15, tile 2g+1
// rows wm + 32g + 2*l15 + 1 — so that the two fragments a lane needs for a tile pair are ADJACENT in the M-/N-contiguous
// LDS image and arrive with one 16-byte (f64) / 8-byte (f32) ds_read instead of two reads: half the LDS instructions for
// the same bytes. The C tile is un-permuted in the epilogue. (A K-contiguous A image has no adjacent rows: A stays
// unpaired there.)
// One C tile (tx, ty) by the calling workgroup; `lds` is the kernel's ONE __shared__ object of gl_lds_elems<T, TM, TN, BK, NST>()
// elements (the body is force-inlined into the kernel that declares it, so every LDS address stays a known LDS address).
template <typename T, int TM, int TN, int BK, int NST>
constexpr int gl_lds_elems() { return NST * (TM * BK + TN * BK); }

template <typename T, typename CA, typename CB, bool BETA0, bool AK, int TM, int TN, int WM, int WN, int BK, int NST,
          bool SPREAD = true, bool PIN = true, bool PAIR = false, int PFD = 1, bool SWAPC = false, bool NTC = false, bool XNOBAR = false, bool UNR = false>
__device__ __forceinline__ void
gl_gemm_tile(T *lds, const int tx, const int ty, T *__restrict__ C, int64_t ldc, const T *__restrict__ A, int64_t lda,
             const T *__restrict__ B, int64_t ldb, GlShape S, CA alpha, CB beta) {
  constexpr int NW = WM * WN;
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int EPI = 1024 / (int)sizeof(T);          // elements per DMA wave-instruction
  constexpr int AIMG = TM * BK, BIMG = TN * BK;       // elements per stage image
  constexpr int NIA = AIMG / EPI, NIB = BIMG / EPI, NI = NIA + NIB;
  static_assert(AIMG % EPI == 0 && BIMG % EPI == 0, "stage images must be whole DMA instructions");
  constexpr int PW = (NI + NW - 1) / NW;              // DMA instructions per wave per stage
  constexpr int MT = TM / WM / 16, NT = TN / WN / 16;
  static_assert(MT >= 1 && NT >= 1 && MT * 16 * WM == TM && NT * 16 * WN == TN, "bad wave layout");
  static_assert(BK % 8 == 0 && NST >= 3 && NST <= 4, "bad pipeline shape");
  static_assert(NI % NW == 0, "every wave must issue the same number of DMA instructions per stage (counted vmcnt)");
  static_assert(gl_lds_elems<T, TM, TN, BK, NST>() == NST * (AIMG + BIMG), "LDS size helper out of step");
  // which operand's tiles are paired, and the XOR swizzle (in elements, applied on odd k-rows) of an M-/N-contiguous
  // image s[k][TX]. It has to suit the ds_read that fetches the fragments (MI355X_MICROARCH.md, LDS table): the 16
  // lanes of one k-row read 16 consecutive fragments, and the lanes a service group takes from two adjacent k-rows
  // must land on different banks.
  //   ds_read_b32 (f32):        32-lane groups, 128-B bank window, 16 lanes cover  64 B -> odd rows shifted by  64 B
  //   ds_read_b64 (f64):        32-lane groups, 256-B bank window, 16 lanes cover 128 B -> odd rows shifted by 128 B
  //   ds_read_b64 (f32 pairs):  as above                                                -> odd rows shifted by 128 B
  //   ds_read_b128 (f64 pairs): 16-lane groups {0-3,12-15,20-27}.. take COMPLEMENTARY lanes of the two rows, which
  //                             already cover 256 B between them -> no shift (a 128-B shift makes them collide 2-way:
  //                             SQ_LDS_BANK_CONFLICT, profiles/r03_pmc_gemm_pair.txt)
  constexpr bool PA = PAIR && !AK && (TM / WM / 16) % 2 == 0, PB = PAIR && (TN / WN / 16) % 2 == 0;
  constexpr int SWA = !PA ? 16 : (sizeof(T) == 8 ? 0 : 32), SWB = !PB ? 16 : (sizeof(T) == 8 ? 0 : 32);
  static_assert(SWA < TM && SWB < TN, "swizzle must stay inside an image row");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bm = tx * TM, bn = ty * TN;
  const int wm = (wave % WM) * (TM / WM), wn = (wave / WM) * (TN / WN);
  const int M = S.M, N = S.N, K = S.K;

  // ---- DMA source addressing: uniform 64-bit base (SGPR, advanced per slab with scalar ALU only) + a per-lane
  // 32-bit byte offset that never changes — `global_load_lds_dwordx4 v_off, s[base:base+1]`. (Per-lane 64-bit
  // address arithmetic on the vector ALU for every DMA instruction costs the matrix pipe ~20 % of its issue slots:
  // MFMA and the other VALU instructions of the 4 waves on a SIMD share one issue port.)
  // instruction q of a stage: q < NIA -> A image chunk q, else B image chunk q - NIA.
  const char *tileA = reinterpret_cast<const char *>(AK ? A + (int64_t)bm * lda : A + bm);   // uniform
  const char *tileB = reinterpret_cast<const char *>(B + bn);
  const int64_t slabA = (AK ? (int64_t)1 : lda) * (int64_t)sizeof(T);    // bytes per unit of k (uniform)
  const int64_t slabB = ldb * (int64_t)sizeof(T);
  uint32_t voff[PW];     // per-lane byte offset from the tile base for k0 = 0
  int krow[PW];          // k (within the slab) this lane reads: only the K-tail clamp needs it
  int64_t rowoff[PW];    // the k-independent part of voff, for the clamped form
#pragma unroll
  for (int p = 0; p < PW; ++p) {
    const int q = wave + p * NW;
    const bool isA = q < NIA;
    const int flat = (isA ? q : q - NIA) * EPI + lane * VEC;
    if (isA && AK) {
      // image sA[i][BK], element (i,k) stored at i*BK + (k ^ g(i))
      const int i = flat / BK, ks = flat % BK;
      const int g = ((BK >= 32 ? (i & 15) : ((i >> 1) & 7)) * 2) & (BK - 1) & ~(VEC - 1);
      const int k = ks ^ g;
      int ri = i;
      if (bm + ri > M - 1) ri = M - 1 - bm;            // clamp to a valid (unused) row
      rowoff[p] = (int64_t)ri * lda * (int64_t)sizeof(T);
      krow[p] = k;
      voff[p] = (uint32_t)(rowoff[p] + (int64_t)k * (int64_t)sizeof(T));
    } else {
      // image s[k][TX], element (k,c) stored at k*TX + (c ^ (k odd ? SW : 0))
      const int TX = isA ? TM : TN;
      const int k = flat / TX, cs = flat % TX;
      int c = cs ^ ((k & 1) ? (isA ? SWA : SWB) : 0);
      const int lim = (isA ? M - bm : N - bn) - VEC;   // >= 0: M, N and the tile origin are multiples of VEC
      if (c > lim) c = lim;                            // clamp to a valid (unused) column
      rowoff[p] = (int64_t)c * (int64_t)sizeof(T);
      krow[p] = k;
      voff[p] = (uint32_t)(rowoff[p] + (int64_t)k * (isA ? lda : ldb) * (int64_t)sizeof(T));
    }
  }
  // CLAMP = false: the slab lies wholly inside K (no per-lane work at all). CLAMP = true: the K tail — an M-/N-
  // contiguous operand reads row min(k, K-1), a K-contiguous A reads 16 B along k and clamps the vector start
  // (K % VEC == 0 there); the slack rows are zeroed in LDS afterwards.
  auto issue_one = [&]<bool CLAMP>(int p, int stage, int k0) {
    const int q = wave + p * NW;      // < NI always (NI % NW == 0); uniform
    const bool isA = q < NIA;
    const char *sb = (isA ? tileA : tileB) + (int64_t)k0 * (isA ? slabA : slabB);
    T *l = lds + stage * (AIMG + BIMG) + q * EPI;
    const char *g;
    if constexpr (!CLAMP) {
      // keep "uniform base + zero-extended 32-bit lane offset" TOGETHER at the load: that is the pattern the backend turns
      // into `global_load_lds_dwordx4 v_off, s[base:base+1]`. Left alone, the optimiser hoists the loop-invariant
      // (tile + lane offset) out of the K loop as a 64-bit per-lane pointer and re-adds the slab advance with one
      // v_lshl_add_u64 per DMA instruction — 6 vector-ALU instructions per slab between the MFMAs. The empty asm makes
      // the lane offset opaque at every use, so there is nothing loop-invariant to hoist.
      asm volatile("" : "+v"(voff[p]));   // (in place: no copy)
      g = sb + voff[p];
    } else {
      int k = k0 + krow[p];
      const int kmax = (AK && isA) ? K - VEC : K - 1;
      k = k < kmax ? k : kmax;
      g = (isA ? tileA : tileB) + rowoff[p] + (int64_t)k * (isA ? slabA : slabB);
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)l, 16, 0, 0);
  };
  auto issue = [&]<bool CLAMP>(int stage, int k0) {
#pragma unroll
    for (int p = 0; p < PW; ++p) issue_one.template operator()<CLAMP>(p, stage, k0);
  };

  using Acc = typename GlMfma<T>::Acc;
  Acc acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[a][b][r] = 0;

  const int l15 = lane & 15, l4 = lane >> 4;
  // fragment offsets inside a stage image for k-step 0 (k = l4); a k-step advances k by 4, which keeps k & 1 and
  // (for the K-contiguous image) the XOR pattern's low bits: offsets for step ks follow by adding a constant.
  typedef T Pair2 __attribute__((ext_vector_type(2)));
  int offA[MT], offB[NT];
#pragma unroll
  for (int a = 0; a < MT; ++a) {
    const int i = PA ? wm + (a >> 1) * 32 + 2 * l15 + (a & 1) : wm + a * 16 + l15;
    if constexpr (AK) {
      const int g = ((BK >= 32 ? (i & 15) : ((i >> 1) & 7)) * 2) & (BK - 1) & ~(VEC - 1);
      offA[a] = i * BK + g;                    // element (i,k) at i*BK + (k ^ g): k is XORed per step below
    } else {
      offA[a] = l4 * TM + (i ^ ((l4 & 1) ? SWA : 0));
    }
  }
#pragma unroll
  for (int b = 0; b < NT; ++b) {
    const int j = PB ? wn + (b >> 1) * 32 + 2 * l15 + (b & 1) : wn + b * 16 + l15;
    offB[b] = AIMG + l4 * TN + (j ^ ((l4 & 1) ? SWB : 0));
  }
  auto frag = [&](const T *st, int ks, T (&av)[MT], T (&bv)[NT]) {
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      if constexpr (AK) {
        const int i = wm + a * 16 + l15;
        const int g = ((BK >= 32 ? (i & 15) : ((i >> 1) & 7)) * 2) & (BK - 1) & ~(VEC - 1);
        av[a] = st[i * BK + ((ks * 4 + l4) ^ g)];
      } else if constexpr (PA) {
        if ((a & 1) == 0) {
          const Pair2 v2 = *reinterpret_cast<const Pair2 *>(st + offA[a] + ks * 4 * TM);
          av[a] = v2[0];
          av[a + 1] = v2[1];
        }
      } else {
        av[a] = st[offA[a] + ks * 4 * TM];
      }
    }
#pragma unroll
    for (int b = 0; b < NT; ++b) {
      if constexpr (PB) {
        if ((b & 1) == 0) {
          const Pair2 v2 = *reinterpret_cast<const Pair2 *>(st + offB[b] + ks * 4 * TN);
          bv[b] = v2[0];
          bv[b + 1] = v2[1];
        }
      } else {
        bv[b] = st[offB[b] + ks * 4 * TN];
      }
    }
  };
  auto compute_tail = [&](int stage, int ksteps) {   // last, partial slab: ksteps < BK/4 (runtime)
    const T *st = lds + stage * (AIMG + BIMG);
    for (int ks = 0; ks < ksteps; ++ks) {
      T av[MT], bv[NT];
      frag(st, ks, av, bv);
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = SWAPC ? GlMfma<T>::run(bv[b], av[a], acc[a][b]) : GlMfma<T>::run(av[a], bv[b], acc[a][b]);
    }
  };

  const int nk = (K + BK - 1) / BK;          // slabs, the last one possibly partial
  const int nfull = K / BK;                  // full slabs
  constexpr int KS = BK / 4;                 // 4-deep MFMA steps per full slab
  static_assert(KS % 2 == 0, "the fragment double buffer relies on an even step count per slab");
  auto next_stage = [&](int st) { return st + 1 == NST ? 0 : st + 1; };
  auto prev_stage = [&](int st) { return st == 0 ? NST - 1 : st - 1; };
  // ---- schedule (slab s lives in ring stage s % NST):
  //   prologue            : issue slabs 0 .. NST-2, wait for slab 0, barrier
  //   slab `it`, step KS/2: wait for slab it+1 (slabs it+2 .. may stay in flight), barrier — this publishes slab
  //                         it+1 AND proves every wave is past slab it-1 — then refill the stage of slab it-1
  //                         with slab it+NST-1, one DMA instruction behind each remaining step's MFMAs
  //   slab `it`, last step: the fragments of step 0 of slab it+1 are read (already published), so the MFMA stream
  //                         runs across the slab boundary: neither the barrier nor the LDS read latency sits
  //                         between the last MFMA of one slab and the first of the next.
#pragma unroll
  for (int s = 0; s < NST - 1; ++s) {
    if (s < nfull) issue.template operator()<false>(s, s * BK);
    else if (s < nk) issue.template operator()<true>(s, s * BK);
  }
  {
    const int ahead = nk - 1;                // slabs issued after slab 0 (capped at NST-2 by the prologue)
    if (ahead >= NST - 2) gl_wait_vmcnt<PW *(NST - 2)>();
    else if (NST > 3 && ahead == 1) gl_wait_vmcnt<PW>();
    else gl_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
  }
  // PFD: how many k-steps ahead a step's fragments are read from LDS (ring of 2*PFD register buffers). One step hides
  // the LDS latency behind the other wave of the SIMD; a workgroup with ONE wave per SIMD (32x32 tiles) needs two.
  constexpr int NFB = 2 * PFD;
  static_assert((PFD == 1 || PFD == 2) && KS % NFB == 0, "fragment ring must divide the steps of a slab");
  T av[NFB][MT], bv[NFB][NT];
  int stage = 0;
  // one full slab. MORE: a slab it+1 exists; NEXTF: it is a full slab (cross-slab fragment prefetch);
  // REFILL: slab it+NST-1 exists (1: a full slab, 2: the K tail, clamped addressing); AHEAD2: slab it+2 exists and was issued earlier (NST == 4 only: it may stay in flight)
  // STG >= 0: the ring index of this slab is a compile-time constant (steady state, UNR): every LDS address of the slab —
  // fragment reads, DMA destinations — is then a loop-invariant base plus an immediate, no vector-ALU address work
  auto slab = [&]<bool MORE, bool NEXTF, int REFILL, bool AHEAD2, int STG = -1>(int it) {
    const int cur = STG >= 0 ? STG : stage;
    const T *st = lds + cur * (AIMG + BIMG);
    const int nst = next_stage(cur);
    const T *stn = lds + nst * (AIMG + BIMG);
    const int rst = prev_stage(cur);
    const int k0n = (it + NST - 1) * BK;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks + PFD < KS) frag(st, ks + PFD, av[(ks + PFD) % NFB], bv[(ks + PFD) % NFB]);
      else if constexpr (NEXTF) frag(stn, ks + PFD - KS, av[(ks + PFD) % NFB], bv[(ks + PFD) % NFB]);
      if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);   // reads first: left alone, hipcc sinks them behind the MFMAs
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
          acc[a][b] = SWAPC ? GlMfma<T>::run(bv[ks % NFB][b], av[ks % NFB][a], acc[a][b])
                            : GlMfma<T>::run(av[ks % NFB][a], bv[ks % NFB][b], acc[a][b]);
      if (ks == KS / 2 - 1) {
        if constexpr (MORE) {
          if constexpr (NST == 4 && AHEAD2) gl_wait_vmcnt<PW>();
          else gl_wait_vmcnt<0>();
          if constexpr (!XNOBAR) __builtin_amdgcn_s_barrier();   // XNOBAR: timing experiment only (results are wrong)
        }
        if constexpr (REFILL != 0 && !SPREAD) issue.template operator()<REFILL == 2>(rst, k0n);
      }
      if constexpr (REFILL != 0 && SPREAD) {
        if (ks >= KS / 2) {
#pragma unroll
          for (int p = 0; p < PW; ++p)
            if (KS / 2 + p * (KS / 2) / PW == ks) issue_one.template operator()<REFILL == 2>(p, rst, k0n);
        }
      }
      if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
    }
    stage = nst;
  };
  if (nfull > 0) {
#pragma unroll
    for (int p = 0; p < PFD; ++p) frag(lds, p, av[p], bv[p]);
  }
  int it = 0;
  if constexpr (UNR) {   // steady state, NST slabs per trip: slab it + j lives in ring stage j (it is a multiple of NST)
    for (; it + 2 * NST - 2 < nfull; it += NST) {
      [&]<int... J>(std::integer_sequence<int, J...>) {
        (slab.template operator()<true, true, 1, true, J>(it + J), ...);
      }(std::make_integer_sequence<int, NST>{});
    }
  }
  for (; it + NST - 1 < nfull; ++it) slab.template operator()<true, true, 1, true>(it);   // steady state
  for (; it < nfull; ++it) {                                                             // last slabs
    const bool more = it + 1 < nk, nextf = it + 1 < nfull, refill = it + NST - 1 < nk, ahead2 = it + 2 < nk;
    if (refill) {            // it + NST - 1 == nfull here: the refilled slab is the K tail; => more, ahead2
      if (nextf) slab.template operator()<true, true, 2, true>(it);
      else slab.template operator()<true, false, 2, true>(it);
    } else if (more) {
      if (nextf) {
        if (ahead2) slab.template operator()<true, true, 0, true>(it);
        else slab.template operator()<true, true, 0, false>(it);
      } else {
        if (ahead2) slab.template operator()<true, false, 0, true>(it);
        else slab.template operator()<true, false, 0, false>(it);
      }
    } else {
      slab.template operator()<false, false, 0, false>(it);
    }
  }
  if (nfull < nk) {                            // K tail: 1 .. BK-1 valid k-rows; published by the barrier above
    const int krem = K - nfull * BK, ksteps = (krem + 3) >> 2;
    if (krem & 3) {                            // zero the slack k-rows of the last 4-deep step (both images)
      T *sA = lds + stage * (AIMG + BIMG), *sB = sA + AIMG;
      const int kz0 = krem, kz1 = ksteps * 4;
      for (int idx = tid; idx < (kz1 - kz0) * TM; idx += NW * 64) {
        const int k = kz0 + idx / TM, i = idx % TM;
        if constexpr (AK) {
          const int g = ((BK >= 32 ? (i & 15) : ((i >> 1) & 7)) * 2) & (BK - 1) & ~(VEC - 1);
          sA[i * BK + (k ^ g)] = 0;
        } else {
          sA[k * TM + i] = 0;
        }
      }
      for (int idx = tid; idx < (kz1 - kz0) * TN; idx += NW * 64) sB[(kz0 + idx / TN) * TN + idx % TN] = 0;
      __syncthreads();
    }
    compute_tail(stage, ksteps);
  }

  // ---- epilogue: res = (alpha*acc) (+ beta*C), each product rounded separately (no FMA)
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // SWAPC: the MFMA ran with its operands exchanged, so the accumulator holds the TRANSPOSED 16x16 tile — the lane
        // index (l15) runs along M, the register index along N: for a column-major C the 16 lanes of a quarter-wave
        // then store 16 consecutive rows (one whole 128-byte line in f64) instead of four rows of 16 different columns
        const int im = SWAPC ? l15 : GlMfma<T>::row(lane, r), jn = SWAPC ? GlMfma<T>::row(lane, r) : l15;
        const int gi = PA ? bm + wm + (a >> 1) * 32 + 2 * im + (a & 1) : bm + wm + a * 16 + im;
        const int gj = PB ? bn + wn + (b >> 1) * 32 + 2 * jn + (b & 1) : bn + wn + b * 16 + jn;
        if (gi < M && gj < N) {
          T *p = C + gi + (int64_t)gj * ldc;
          const CA t = alpha * (CA)acc[a][b][r];
          T o;
          if constexpr (BETA0) o = (T)t;
          else {   // α-term in α's type, β-term in β's, summed in the wider (see stream_kernels.h: fin_ab)
            using P = std::conditional_t<(sizeof(CA) >= sizeof(CB)), CA, CB>;
            o = (T)((P)t + (P)(beta * (CB)(*p)));
          }
          // NTC: streaming stores — the tile is not re-read by this kernel, and lines that do not stay dirty in L2 do
          // not have to be written back at the kernel boundary the dependent GEMM waits behind
          if constexpr (NTC) __builtin_nontemporal_store(o, p);
          else *p = o;
        }
      }
}

template <typename T, typename CA, typename CB, bool BETA0, bool AK, int TM, int TN, int WM, int WN, int BK, int NST,
          bool SPREAD = true, bool PIN = true, bool PAIR = false, int PFD = 1, bool SWAPC = false, bool NTC = false, bool XNOBAR = false, bool UNR = false>
__global__ void __launch_bounds__(WM * WN * 64)
gemm_glds_kernel(T *__restrict__ C, int64_t ldc, const T *__restrict__ A, int64_t lda,
                 const T *__restrict__ B, int64_t ldb, GlShape S, CA alpha, CB beta) {
  __shared__ __attribute__((aligned(1024))) T lds[gl_lds_elems<T, TM, TN, BK, NST>()];   // ONE LDS object (a second one makes
                                                                                         // hipcc drain vmcnt before every ds_read)
  int tx, ty;
  gl_tile_of(blockIdx.x, S.gx, S.gy, tx, ty);
  gl_gemm_tile<T, CA, CB, BETA0, AK, TM, TN, WM, WN, BK, NST, SPREAD, PIN, PAIR, PFD, SWAPC, NTC, XNOBAR, UNR>(lds, tx, ty, C, ldc, A, lda, B, ldb, S,
                                                                                                          alpha, beta);
}

// ---- kron = two DEPENDENT GEMMs in ONE launch, the dependency kept inside an XCD (round 6) --------------------------------
//   phase 1: Ut (m x q) = opA * X^T          tile (r, ty): row block r of Ut
//   phase 2: R  (p x m) = opB * Ut^T          tile (ti, r): column block r of R needs row block r of Ut — ALL of it —
// i.e. a barrier between the two products, which as two launches costs the launch boundary (2.2 us for ANY dependent
// kernel) plus a cold DMA ring per product, and as a device-wide barrier inside one launch costs as much (agent-scope
// release fences walk the L2; the fence-free form is three trips through the memory-side fabric). Here the dependency
// never leaves an XCD: row block r is produced AND consumed by the workgroups whose id is r modulo 8 — workgroups are dealt
// to the XCDs round-robin, so these share ONE XCD (which one varies from launch to launch; the period is probed once per
// ctx against the hardware's XCC_ID register: dense.hip xcd_map_ok). The producers' tiles sit in that XCD's L2, the counter that publishes them is an atomic
// executed in that same L2 (workgroup-scope atomics: no sc1, no write-back, no invalidate), and the consumers read both
// from there: no agent-scope fence, no trip to the memory side. The wait is bounded (ctx fault word, as every single-launch
// form of the library); the grid must be co-resident (checked by the launch site).
// Counters: cnt[r] counts the finished phase-1 tiles of row block r, done[r] the phase-2 workgroups that have seen it
// complete; the last of those re-arms both (self-cleaning: a launch leaves every counter at zero).
constexpr int kGlFuseStride = 32;
struct GlFuse {
  unsigned *cnt;            // [2 * nrb] counters, kGlFuseStride words apart (one 128-byte line each): cnt[r], done[r]
  int nrb;                  // row blocks of Ut = column blocks of R
  int gy1;                  // phase-1 tiles per row block (producers)
  int gx2;                  // phase-2 tiles per column block (consumers)
  int per_xcd;              // workgroups per XCD = grid / 8
  unsigned long long ticks; // bounded wait (wall_clock64 ticks)
  unsigned *fault;          // ctx fault word (pinned host memory)
  unsigned fault_code;
  int drop;                 // TEST HOOK (tune key fused_debug_drop): the producer with this workgroup id never signals
};

__device__ __forceinline__ unsigned gl_xcc_id() {
  return (unsigned)__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 15u;   // HW_REG_XCC_ID[3:0]
}

template <typename T, typename CA, typename CB, bool BETA0, bool AK, int TM, int TN, int WM, int WN, int BK, int NST,
          bool PAIR, int PFD, bool UNR>
__global__ void __launch_bounds__(WM * WN * 64)
kron_fused_kernel(T *__restrict__ R, int64_t ldr, const T *__restrict__ Bf, int64_t ldb, T *__restrict__ Ut,
                  int64_t ldu, const T *__restrict__ Af, int64_t lda, const T *__restrict__ X, int64_t ldx, GlShape S1,
                  GlShape S2, CA alpha, CB beta, GlFuse F) {
  static_assert(TM == TN, "a row block of the first product is a column block of the second");
  __shared__ __attribute__((aligned(1024))) T lds[gl_lds_elems<T, TM, TN, BK, NST>()];
  const int tid = threadIdx.x;
  // `xcd` is a LABEL: workgroups whose ids agree modulo 8 share a physical XCD (round-robin dispatch; the launch site has
  // probed the period), whichever one that is for this launch
  const int xcd = (int)(blockIdx.x & 7u), local = (int)(blockIdx.x >> 3);
  // row blocks of this XCD: r = xcd + 8 * t, t < nmine
  const int nmine = F.nrb > xcd ? (F.nrb - xcd + 7) / 8 : 0;
  // ---- phase 1: tile (r, ty) of Ut
  if (local < nmine * F.gy1) {
    const int r = xcd + 8 * (local / F.gy1), ty = local % F.gy1;
    gl_gemm_tile<T, double, double, true, AK, TM, TN, WM, WN, BK, NST, true, true, PAIR, PFD, true, false, false, UNR>(
        lds, r, ty, Ut, ldu, Af, lda, X, ldx, S1, 1.0, 0.0);
    __builtin_amdgcn_s_waitcnt(0);                  // this wave's stores have reached the XCD's L2 ...
    __syncthreads();                                 // ... and so have every wave's
    if (tid == 0 && (int)blockIdx.x != F.drop)
      __hip_atomic_fetch_add(F.cnt + (size_t)r * kGlFuseStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // executed in that L2
  }
  // ---- phase 2: tile (ti, r) of R
  if (local < nmine * F.gx2) {
    const int r = xcd + 8 * (local / F.gx2), ti = local % F.gx2;
    if (tid == 0) {                                  // (no second __shared__ object in this kernel: see gemm_glds_kernel)
      lds[0] = T(0);                                 // "timed out" flag for the workgroup (the ring is idle here)
      unsigned long long t0 = 0;
      unsigned it = 0;
      // the poll must be an atomic EXECUTED IN L2 on every trip: an idempotent read-modify-write (fetch_add 0) is folded into
      // a plain load by the compiler, which the CU's L1 then serves from its first, stale copy for ever; a compare-and-swap
      // of the target value with itself is not
      auto peek = [&]() {
        unsigned expect = (unsigned)F.gy1;
        __hip_atomic_compare_exchange_strong(F.cnt + (size_t)r * kGlFuseStride, &expect, (unsigned)F.gy1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return expect;                               // the value found
      };
      while (F.ticks != 0 && peek() < (unsigned)F.gy1) {   // (ticks == 0: timing experiment kron_fuse = 2 — no wait, WRONG results)
        __builtin_amdgcn_s_sleep(8);                 // ~0.2 us between polls: the producers' atomics are not queued behind pollers
        if ((++it & 255u) == 0) {
          const unsigned long long now = (unsigned long long)wall_clock64();
          if (t0 == 0) t0 = now;
          else if (now - t0 > F.ticks) {
            __hip_atomic_store(F.fault, F.fault_code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            lds[0] = T(1);
            break;                                   // reported through the fault word; this tile is stored as NaN below
          }
        }
      }
      // the last consumer of row block r re-arms its counters for the next launch
      if (__hip_atomic_fetch_add(F.cnt + (size_t)(F.nrb + r) * kGlFuseStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == (unsigned)F.gx2 - 1u) {
        __hip_atomic_store(F.cnt + (size_t)r * kGlFuseStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(F.cnt + (size_t)(F.nrb + r) * kGlFuseStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    __syncthreads();
    const bool timed_out = lds[0] != T(0);
    __syncthreads();                                 // everybody has read the flag before the ring is filled
    if (timed_out) {                                 // the row block never became complete: NaN, like every single-launch form
      for (int idx = tid; idx < TM * TN; idx += WM * WN * 64) {
        const int64_t gi = (int64_t)ti * TM + idx % TM, gj = (int64_t)r * TN + idx / TM;
        if (gi < S2.M && gj < S2.N) R[gi + gj * ldr] = T(__builtin_nan(""));
      }
      return;
    }
    gl_gemm_tile<T, CA, CB, BETA0, AK, TM, TN, WM, WN, BK, NST, true, true, PAIR, PFD, true, false, false, UNR>(
        lds, ti, r, R, ldr, Bf, ldb, Ut, ldu, S2, alpha, beta);
  }
}

// Preconditions of the DMA path (16-byte global reads): pointers 16-byte aligned, leading dimensions and the
// contiguous extents (M for an M-contiguous A, K for a K-contiguous A, N for B') multiples of the vector width.
template <typename T>
inline bool gemm_glds_ok(const T *A, int64_t lda, bool a_kcontig, const T *B, int64_t ldb, int64_t M, int64_t N,
                         int64_t K) {
  constexpr int VEC = 16 / (int)sizeof(T);
  if (M < VEC || N < VEC || K < (a_kcontig ? VEC : 1)) return false;
  if ((((uintptr_t)A | (uintptr_t)B) & 15u) != 0 || lda % VEC || ldb % VEC || N % VEC) return false;
  return a_kcontig ? (K % VEC == 0) : (M % VEC == 0);
}

}  // namespace mxlo
