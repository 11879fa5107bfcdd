// helicon_hip.hip — MI355X (gfx950 / CDNA4) kernels + C ABI for the denovo3D (twist, rise, Csym)
// sweep.  See include/helicon_hip.h for the boundary and DESIGN.md for the data layout.
//
// Three pipelines give the same scores; reference semantics in brackets:
//   any candidate list ("transform"):
//     k_first_pass    rasterise the helical lattice of Gaussian balls straight into the FFT input
//                     registers [utils.py:91-106, 153-172], FFT every image COLUMN (along y) with
//                     two real columns packed into one complex transform, write the half spectrum
//                     H (N/2 x N complex, row 0 packs ky=0 and ky=N/2) in 128-byte lines of
//                     8 ky x 1 column pair.
//     k_second_pass   transpose 8-row blocks of H through LDS, FFT every row along x ->
//                     F[ky][kx] on the half plane, then a=|F|, q=log1p(a) [transforms.py:805-810]
//                     and the three masked moments sum w q, sum w q^2, sum w (E-Ebar) q with
//                     Hermitian weights w in {0,1,2}; one partial triple per spectrum row.
//   runs of candidates sharing (twist, csym, rot), no tilt/psi — the driver's twist-major grid:
//     k_run_table     column transforms of every subunit's footprint, once per run.
//     k_first_pass_table + k_second_pass ("run tables"): H as a short weighted sum of table rows.
//     k_fused_pass ("fused", the default): one ky block of up to 64 candidates per workgroup, H built
//                     in LDS from the table slice and the candidate's column factors, row FFT and
//                     moments in place; no intermediate in HBM.
//   finalize          Pearson coefficient from the moments [analysis.py:793-799]: grid layer 0 of the
//                     next batch's first kernel, or k_finalize.
//   several segments: the second / fused pass keeps q, k_segment_corr contracts it with all segments'
//                     centred spectra on the f32 matrix cores, k_finalize_segments scores S x G.
// Everything is wave64 code: an N-point FFT is owned by N/8 lanes holding 8 points each
// (one wavefront for N = 512), Stockham radix-8/4/2 stages exchange through LDS.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <system_error>
#include <thread>
#include <unordered_set>
#include <utility>
#include <vector>

#include "../../include/helicon_hip.h"
#include "gen_rows.h"  // the general-size sweep's two-step row kernel (second translation unit, gen_rows.hip)

// Tuning knobs (compile-time; defaults are the measured best)
#ifndef HH_ABLATE
#define HH_ABLATE 0        // timing-only builds: 1 no raster, 2 no K_A FFT, 4 no K_A store, 8 no twiddle loads,
                           // 16 no K_B FFT, 32 no K_B epilogue math, 64 no K_B weight loads; run-table first pass:
                           // 128 no stores, 256 no accumulation, 1024 no table staging; fused pass: 2048 no panel
                           // accumulation, 4096 no column-factor prefetch (16 / 32 as for K_B), 8192 no second exchange,
                           // 65536 compact q stores of every candidate into 64 rows (no HBM write stream)
#endif
#ifndef HH_KA_WPS
#define HH_KA_WPS 8        // K_A: waves per SIMD the register allocator must leave room for (4 workgroups per CU)
#endif
#ifndef HH_KB_TWLDS
#define HH_KB_TWLDS 0      // K_B: twiddles from a per-workgroup LDS table instead of registers
#endif
#ifndef HH_KA_FPW_BIG
#define HH_KA_FPW_BIG 4      // K_A: transforms per workgroup and tile when a transform spans two wavefronts (N = 1024)
#endif
#ifndef HH_KA_BAND
#define HH_KA_BAND 64       // K_A: image columns per workgroup (N >= 256)
#endif
#ifndef HH_KB_CPW
#define HH_KB_CPW 16       // K_B: candidates per workgroup (one ky block of 8 rows of each)
#endif
#ifndef HH_KT_PAIRS
#define HH_KT_PAIRS 32     // run-table first pass: column pairs per workgroup
#endif
#ifndef HH_KT_KYW
#define HH_KT_KYW 128      // run-table first pass: ky rows per workgroup
#endif
#ifndef HH_KF_CPW
#define HH_KF_CPW 0        // fused pass: candidates per workgroup (of one run); 0 = chosen per launch (fused_groups_per_run)
#endif
#ifndef HH_FUSED_BATCH
#define HH_FUSED_BATCH 131072  // fused pass: most candidates per launch (whole runs); 6 GB of column factors + moments at N = 512
#endif
#ifndef HH_FUSED_BYTES
#define HH_FUSED_BYTES (8LL << 30)  // fused pass: the launch is shortened so that its column factors (two halves) fit this
#endif
#ifndef HH_SEG_BATCH
#define HH_SEG_BATCH 24576  // fused pass with several segments: most candidates per launch (C5: 11.6 ms at 1024, 9.5 ms at 20k)
#endif
#ifndef HH_SEG_BYTES
#define HH_SEG_BYTES (12LL << 30)  // ... shortened so that the batch's masked q (0.5 MB per candidate at N = 512) fits this
#endif
#ifndef HH_KF_WPS
#define HH_KF_WPS 4        // fused pass: waves per SIMD the register allocator must leave room for
#endif
#ifndef HH_FFT_SWZ
#define HH_FFT_SWZ 1       // xor-swizzled slots for the first exchange of every transform (bank conflicts)
#endif
#ifndef HH_FFT_SWZ2
#define HH_FFT_SWZ2 1      // second exchange of the 512 / 1024-point transforms kept in bank-conflict-free slots
#endif
#ifndef HH_KF_PSWZ
#define HH_KF_PSWZ 1       // fused pass: 16-byte chunks of the panel row xor-swizzled (conflict-free b128 stores)
#endif
#ifndef HH_KF_STAGGER
#define HH_KF_STAGGER 1    // fused pass (N = 512): wavefronts 4-7 run half a candidate behind wavefronts 0-3
#endif
#ifndef HH_KF_DEFER_Q
#define HH_KF_DEFER_Q 1    // fused pass, several segments: the early wavefronts store a candidate's q at the top of the next round
#endif
#ifndef HH_KF_EGLOBAL
#define HH_KF_EGLOBAL 0    // fused pass (N = 512): column factors read straight from global memory (vector L1) instead of LDS copies
#endif
#ifndef HH_KF_CUT
#define HH_KF_CUT 1        // fused pass: part A of a candidate ends after the butterflies of this transform stage
#endif
#ifndef HH_KF_SPLIT
#define HH_KF_SPLIT 1      // fused pass, N = 1024: one radix-2 step across a row's two wavefronts, then a 512-point
#endif                     // transform inside each (one workgroup barrier per candidate instead of six)
#ifndef HH_KF_PRIO
#define HH_KF_PRIO 0       // fused pass: s_setprio level of the late wavefronts (0 = off)
#endif
#ifndef HH_POISON
#define HH_POISON 0        // test builds: NaN in every factor row the fused pass has no business reading
#endif
#ifndef HH_XCD_MAP
#define HH_XCD_MAP 1       // fused pass: all ky blocks of a layer of candidates on one XCD (shared L2)
#endif
#ifndef HH_KB_WPS
#define HH_KB_WPS 4        // K_B: waves per SIMD the register allocator must leave room for
#endif

namespace {

// ------------------------------------------------------------------------------------------
// small complex helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }  // a * (-i)

// Synchronise the T lanes that own one transform.  Up to 64 lanes live in one wavefront, whose
// LDS instructions execute in program order, so only the compiler has to be fenced; wider groups
// (N = 1024: two wavefronts per transform) need the workgroup barrier.
template <int T>
__device__ __forceinline__ void group_sync() {
  if constexpr (T <= 64) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else {
    __syncthreads();
  }
}

template <int R>
__device__ __forceinline__ void bfly(float2 (&a)[R]);

template <>
__device__ __forceinline__ void bfly<2>(float2 (&a)[2]) {
  float2 t = a[0];
  a[0] = cadd(t, a[1]);
  a[1] = csub(t, a[1]);
}

template <>
__device__ __forceinline__ void bfly<4>(float2 (&a)[4]) {
  float2 s0 = cadd(a[0], a[2]), d0 = csub(a[0], a[2]);
  float2 s1 = cadd(a[1], a[3]), d1 = mul_mi(csub(a[1], a[3]));
  a[0] = cadd(s0, s1);
  a[2] = csub(s0, s1);
  a[1] = cadd(d0, d1);
  a[3] = csub(d0, d1);
}

template <>
__device__ __forceinline__ void bfly<8>(float2 (&a)[8]) {
  constexpr float h = 0.70710678118654752440f;
  float2 e[4] = {cadd(a[0], a[4]), cadd(a[1], a[5]), cadd(a[2], a[6]), cadd(a[3], a[7])};
  float2 o0 = csub(a[0], a[4]), t1 = csub(a[1], a[5]), t2 = csub(a[2], a[6]), t3 = csub(a[3], a[7]);
  float2 o[4] = {o0, make_float2((t1.x + t1.y) * h, (t1.y - t1.x) * h), mul_mi(t2),
                 make_float2((t3.y - t3.x) * h, -(t3.x + t3.y) * h)};
  bfly<4>(e);
  bfly<4>(o);
  a[0] = e[0]; a[1] = o[0]; a[2] = e[1]; a[3] = o[1];
  a[4] = e[2]; a[5] = o[2]; a[6] = e[3]; a[7] = o[3];
}

// ------------------------------------------------------------------------------------------
// FFT plan: radices per N (product = N), 8 points per lane, T = N/8 lanes per transform
// ------------------------------------------------------------------------------------------
template <int N> struct Plan;
template <> struct Plan<32>   { static constexpr int n = 2; static constexpr int r0 = 8, r1 = 4, r2 = 1, r3 = 1; };
template <> struct Plan<64>   { static constexpr int n = 2; static constexpr int r0 = 8, r1 = 8, r2 = 1, r3 = 1; };
template <> struct Plan<128>  { static constexpr int n = 3; static constexpr int r0 = 8, r1 = 8, r2 = 2, r3 = 1; };
template <> struct Plan<256>  { static constexpr int n = 3; static constexpr int r0 = 8, r1 = 8, r2 = 4, r3 = 1; };
template <> struct Plan<512>  { static constexpr int n = 3; static constexpr int r0 = 8, r1 = 8, r2 = 8, r3 = 1; };
template <> struct Plan<1024> { static constexpr int n = 4; static constexpr int r0 = 8, r1 = 8, r2 = 8, r3 = 2; };

constexpr int tw_count(int r) { return r > 1 ? 8 - 8 / r : 0; }  // twiddles a lane needs in a radix-r stage
constexpr int imin(int a, int b) { return a < b ? a : b; }
template <int N> struct TwN {
  using P = Plan<N>;
  static constexpr int T = N / 8;
  // register copy: one slot per (butterfly q, input r) of every twiddled stage
  static constexpr int off1 = 0;
  static constexpr int off2 = off1 + tw_count(P::r1);
  static constexpr int off3 = off2 + tw_count(P::r2);
  static constexpr int total = off3 + tw_count(P::r3) + 1;  // +1: never a zero-length array
  // LDS copy: a stage with NS sub-transform points only has min(NS, T) distinct lane values per
  // slot (the twiddle index is j mod NS), so a slot is that long instead of T
  static constexpr int e1 = imin(P::r0, T), e2 = imin(P::r0 * P::r1, T), e3 = imin(P::r0 * P::r1 * P::r2, T);
  static constexpr int lds1 = 0;
  static constexpr int lds2 = lds1 + tw_count(P::r1) * e1;
  static constexpr int lds3 = lds2 + tw_count(P::r2) * e2;
  static constexpr int lds_total = lds3 + tw_count(P::r3) * e3;  // complex elements
};

// Twiddles depend on the lane only, so a lane fetches them once (from a float64-rounded table
// W_N[k] = exp(-2 pi i k / N)) and keeps them in registers across all its transforms.
// (TS: the table is W_{N TS}, every TS-th entry of it is W_N)
template <int N, int R, int NS, int OFF, int TS = 1>
__device__ __forceinline__ void load_stage_twiddles(float2* tw, int t, const float2* __restrict__ table) {
  constexpr int T = N / 8, NB = 8 / R;
#pragma unroll
  for (int q = 0; q < NB; ++q) {
    const int j = t + q * T;
    const int k = j & (NS - 1);
#pragma unroll
    for (int r = 1; r < R; ++r) tw[OFF + q * (R - 1) + (r - 1)] = table[((r * k * (N / (NS * R))) & (N - 1)) * TS];
  }
}

template <int N, int TS = 1>
__device__ __forceinline__ void load_twiddles(float2 (&tw)[TwN<N>::total], int t, const float2* __restrict__ table) {
  using P = Plan<N>;
  load_stage_twiddles<N, P::r1, P::r0, TwN<N>::off1, TS>(tw, t, table);
  if constexpr (P::n > 2) load_stage_twiddles<N, P::r2, P::r0 * P::r1, TwN<N>::off2, TS>(tw, t, table);
  if constexpr (P::n > 3) load_stage_twiddles<N, P::r3, P::r0 * P::r1 * P::r2, TwN<N>::off3, TS>(tw, t, table);
}

// Where a lane finds its twiddles: its own registers (K_B keeps them across all its rows) or a
// per-workgroup LDS copy (K_A, whose raster needs the registers).  `at<OFF, LOFF, E>(i)` is slot i
// of a stage whose register slots start at OFF and whose LDS slots (E entries each) start at LOFF.
struct TwRegs {
  const float2* p;
  template <int OFF, int LOFF, int E>
  __device__ __forceinline__ float2 at(int i) const { return p[OFF + i]; }
};
struct TwLds {
  const float2* p;
  int t;  // lane index inside the transform
  template <int OFF, int LOFF, int E>
  __device__ __forceinline__ float2 at(int i) const { return p[LOFF + i * E + (t & (E - 1))]; }
};

// Fill the LDS twiddle table (TwN<N>::lds_total entries) from the global W_N table; called by the
// T lanes of one transform group.
template <int N, int R, int NS, int LOFF>
__device__ __forceinline__ void fill_stage_twiddles(float2* lds, int t, const float2* __restrict__ table) {
  constexpr int T = N / 8, NB = 8 / R, E = imin(NS, T);
  if (t < E) {
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const int k = (t + q * T) & (NS - 1);
#pragma unroll
      for (int r = 1; r < R; ++r) lds[LOFF + (q * (R - 1) + (r - 1)) * E + t] = table[(r * k * (N / (NS * R))) & (N - 1)];
    }
  }
}
template <int N>
__device__ __forceinline__ void fill_twiddles_lds(float2* lds, int t, const float2* __restrict__ table) {
  using P = Plan<N>;
  fill_stage_twiddles<N, P::r1, P::r0, TwN<N>::lds1>(lds, t, table);
  if constexpr (P::n > 2) fill_stage_twiddles<N, P::r2, P::r0 * P::r1, TwN<N>::lds2>(lds, t, table);
  if constexpr (P::n > 3) fill_stage_twiddles<N, P::r3, P::r0 * P::r1 * P::r2, TwN<N>::lds3>(lds, t, table);
}

// The exchange buffers are NOT padded (the unpadded buffers are what lets four K_A workgroups, or two
// fused workgroups, share a CU); the first exchange's stride-8 scatter is xor-swizzled instead.
// One Stockham stage.  Lane t owns butterflies j = t + q*T (q < 8/R); butterfly j reads
// in[j + r*N/R] — always the lane's own register slots v[q + r*(8/R)] — and writes
// out[(j/NS)*NS*R + (j mod NS) + r*NS].  The last stage's outputs land back in the same slots,
// so on return v[m] = X[t + m*T].
// PART: 0 = the whole stage; 1 = butterflies and exchange stores only; 2 = exchange loads only (a caller may put other
// work of the same lanes, or a workgroup barrier, between the two halves: the data is in the exchange buffer).
template <int N, int R, int NS, bool LAST, int OFF, int LOFF, bool SWZ1, typename TW, int PART = 0>
__device__ __forceinline__ void fft_stage(float2 (&v)[8], const TW& tw, int t, float2* buf) {
  constexpr int T = N / 8, NB = 8 / R, E = imin(NS, T);
  constexpr bool SWZ = SWZ1 && NS == 1 && R == 8 && !LAST && N >= 128;
  // second exchange (NS = 8, R = 8): lane j writes out[64 (j >> 3) + (j & 7) + 8 r]; lanes j and j + 8 of a 16-lane
  // ds_write_b64 group are 512 bytes apart = the same banks (2-way conflict on all eight stores).  Element o is
  // therefore kept in slot o ^ (((o >> 6) & 1) << 3): the writer's r becomes r ^ b with b = (j >> 3) & 1, i.e. even r
  // go to base + 8 b + 8 r and odd r to base - 8 b + 8 r (two base registers, the immediates stay), and the reader of
  // element n = t + m T looks in n ^ (((n >> 6) & 1) << 3).  No extra vector instructions.
  constexpr bool SWZ2 = SWZ1 && HH_FFT_SWZ2 && NS == 8 && R == 8 && !LAST && NB == 1 && (T == 64 || T == 128);
  if constexpr (PART != 2) {
#pragma unroll
  for (int q = 0; q < NB; ++q) {
    float2 a[R];
#pragma unroll
    for (int r = 0; r < R; ++r) a[r] = v[q + r * NB];
    if constexpr (NS > 1) {
#pragma unroll
      for (int r = 1; r < R; ++r) a[r] = cmul(a[r], tw.template at<OFF, LOFF, E>(q * (R - 1) + (r - 1)));
    }
    bfly<R>(a);
    if constexpr (LAST) {
#pragma unroll
      for (int r = 0; r < R; ++r) v[q + r * NB] = a[r];
    } else if constexpr (SWZ) {
      // first stage (NS = 1, R = 8): lane j's eight outputs go to slots 8 j + (r ^ s), s = (j >> 1) & 7.
      // Plain slots 8 j + r put 16 lanes of a ds_write_b64 group on two bank pairs (8-way conflict);
      // with the xor the 16 lanes hit 16 different pairs.  Byte address = (64 j | 8 s) ^ 8 r.
      const int j = t + q * T;
      const unsigned bs = (unsigned)(j * 64) | (unsigned)(((j >> 1) & 7) << 3);
      char* const base = reinterpret_cast<char*>(buf);
#pragma unroll
      for (int r = 0; r < R; ++r) *reinterpret_cast<float2*>(base + (bs ^ (unsigned)(r << 3))) = a[r];
    } else if constexpr (SWZ2 && (HH_ABLATE & 8192) != 0) {
      // timing-only: the second exchange costs nothing (values stay where they are)
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = a[r];
    } else if constexpr (SWZ2) {
      const int j = t;  // NB == 1
      const int k = j & 7, b8 = ((j >> 3) & 1) * 8;
      float2* const we = buf + (j - k) * 8 + k + b8;  // even r
      float2* const wo = buf + (j - k) * 8 + k - b8;  // odd r
#pragma unroll
      for (int r = 0; r < 8; ++r) ((r & 1) ? wo : we)[r * 8] = a[r];
    } else {
      const int j = t + q * T;
      const int k = j & (NS - 1);
      float2* const w = buf + (j - k) * R + k;  // one base register + immediate offsets r * NS
#pragma unroll
      for (int r = 0; r < R; ++r) w[r * NS] = a[r];
    }
  }
  }  // PART != 2
  if constexpr (!LAST && PART != 1) {
    group_sync<T>();
    if constexpr (SWZ2 && (HH_ABLATE & 8192) != 0) {
    } else if constexpr (SWZ2) {
      if constexpr (T == 64) {  // (n >> 6) & 1 = m & 1
        const int e0 = t, e1 = t ^ 8;
#pragma unroll
        for (int m = 0; m < 8; ++m) v[m] = buf[((m & 1) ? e1 : e0) + m * T];
      } else {                  // T = 128: (n >> 6) & 1 = (t >> 6) & 1 for every m
        const int e0 = t ^ (((t >> 6) & 1) << 3);
#pragma unroll
        for (int m = 0; m < 8; ++m) v[m] = buf[e0 + m * T];
      }
    } else if constexpr (SWZ) {
      // element n = t + m T sits in slot (n & ~7) | ((n & 7) ^ ((n >> 4) & 7))
      if constexpr (T == 64) {  // (n >> 4) & 7 = (t >> 4) ^ 4 (m & 1): two bases, immediate offsets
        const int e0 = (t & ~7) | ((t & 7) ^ (t >> 4)), e1 = e0 ^ 4;
#pragma unroll
        for (int m = 0; m < 8; ++m) v[m] = buf[((m & 1) ? e1 : e0) + m * T];
      } else if constexpr (T == 128) {  // (n >> 4) & 7 = (t >> 4) & 7 for every m
        const int e0 = (t & ~7) | ((t & 7) ^ ((t >> 4) & 7));
#pragma unroll
        for (int m = 0; m < 8; ++m) v[m] = buf[e0 + m * T];
      } else {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          const int nn = t + m * T;
          v[m] = buf[(nn & ~7) | ((nn & 7) ^ ((nn >> 4) & 7))];
        }
      }
    } else {
#pragma unroll
      for (int m = 0; m < 8; ++m) v[m] = buf[t + m * T];
    }
    group_sync<T>();
  }
}

// In: v[m] = x[t + m*T].  Out: v[m] = X[t + m*T], X = forward DFT (exp(-2 pi i nk/N)).
// All T lanes of the transform must call it (group_sync inside; block-wide for T > 64).
// SWZ1: xor-swizzle the first exchange (worth it where LDS time matters more than nine extra vector
// instructions: the second pass and the fused pass; not in the raster + column-transform kernel).
// fft_lanes, optionally cut in two at the exchange after stage CUT (1-based): PHASE 1 runs the stages up to CUT's
// butterflies and leaves the data in the exchange buffer (nothing of the transform lives in registers in between),
// PHASE 2 picks it up there and finishes; PHASE 0 is the whole transform.
template <int N, bool SWZ1, typename TW, int PHASE, int CUT>
__device__ __forceinline__ void fft_lanes_part(float2 (&v)[8], const TW& tw, int t, float2* buf) {
  using P = Plan<N>;
  using W = TwN<N>;
  static_assert(CUT >= 1 && CUT < P::n, "the cut is at an exchange");
  // part of stage s this call runs: 0 whole, 1 front, 2 back, -1 nothing
  auto part = [](int s) constexpr {
    if (PHASE == 0) return 0;
    if (PHASE == 1) return s < CUT ? 0 : (s == CUT ? 1 : -1);
    return s < CUT ? -1 : (s == CUT ? 2 : 0);
  };
  if constexpr (part(1) >= 0) fft_stage<N, P::r0, 1, false, 0, 0, SWZ1, TW, part(1)>(v, tw, t, buf);
  if constexpr (part(2) >= 0)
    fft_stage<N, P::r1, P::r0, P::n == 2, W::off1, W::lds1, SWZ1, TW, part(2)>(v, tw, t, buf);
  if constexpr (P::n > 2 && part(3) >= 0)
    fft_stage<N, P::r2, P::r0 * P::r1, P::n == 3, W::off2, W::lds2, false, TW, part(3)>(v, tw, t, buf);
  if constexpr (P::n > 3 && part(4) >= 0)
    fft_stage<N, P::r3, P::r0 * P::r1 * P::r2, true, W::off3, W::lds3, false, TW, part(4)>(v, tw, t, buf);
}

template <int N, bool SWZ1 = false, typename TW>
__device__ __forceinline__ void fft_lanes(float2 (&v)[8], const TW& tw, int t, float2* buf) {
  fft_lanes_part<N, SWZ1, TW, 0, 1>(v, tw, t, buf);
}

// ------------------------------------------------------------------------------------------
// lattice geometry
// ------------------------------------------------------------------------------------------
struct DevGeom {
  double height;      // nx * apix (utils.py:180)
  double m[6];        // rows 1 (image row axis) and 2 (helical axis) of R_yx(tilt, -psi)
  double dy;
  float apix, inv_apix;
  float inv_sigma2;   // ln2 / ball_radius^2
  int rpx;            // truncation half-window, pixels
  int n_units;
  int has_rot;
  int fast;           // 1: the axial coordinate is monotonic in the subunit index (|m[5]| >= 1/4)
  float slack;        // bound of |axial coordinate - m[5] * i * rise| over the units, Angstrom
  float pad_;
};

struct Cand {  // one candidate, decoded once per workgroup
  double twist, rise, rot;
  int csym, imax, M;
};

__device__ __forceinline__ Cand decode_candidate(const double* __restrict__ p, const DevGeom& g) {
  Cand c;
  c.twist = p[0];
  c.rise = p[1];
  c.csym = (int)p[2];
  c.rot = p[3];
  if (c.csym < 1) c.csym = 1;
  if (c.csym > 360) c.csym = 360;
  double im = ceil(g.height / c.rise);  // utils.py:153
  if (!(im >= 0.0)) im = -1.0;          // rise <= 0 / NaN: empty lattice
  if (im > 1048576.0) im = 1048576.0;
  c.imax = (int)im;
  c.M = c.imax < 0 ? 0 : (2 * c.imax + 1) * c.csym * g.n_units;
  return c;
}

// Centre `ci` of the lattice in Angstrom: (row coordinate, axial coordinate), with the
// reference's arithmetic types (utils.py:138-171): unit position float32, z-rotation by
// twist*i + 360 s/csym in float64 then rounded to float32, axial shift added in float32,
// optional tilt/psi rotation and dy in float64.
__device__ __forceinline__ float2 centre_position(const Cand& c, const DevGeom& g, const double* __restrict__ units,
                                                  int ci) {
  const int u = ci % g.n_units;
  const int is = ci / g.n_units;
  const int s = is % c.csym;
  const int i = is / c.csym - c.imax;
  const double r = units[3 * u], az = units[3 * u + 1];
  const float z = (float)units[3 * u + 2];
  // angles in half-turns for sincospi (exact argument reduction, no large-argument path)
  double s0, c0;
  sincospi(az * 0.31830988618379067 + c.rot * (1.0 / 180.0), &s0, &c0);
  const double c0u = (double)(float)(r * c0), c0v = (double)(float)(r * s0);
  double st, ct;
  sincospi((c.twist * (double)i + (double)s * 360.0 / (double)c.csym) * (1.0 / 180.0), &st, &ct);
  const float cu = (float)(c0u * ct - c0v * st);
  const float cv = (float)(c0u * st + c0v * ct);
  const float ca = z + (float)((double)i * c.rise);
  double yc = cv, xc = ca;
  if (g.has_rot) {
    yc = g.m[0] * (double)cu + g.m[1] * (double)cv + g.m[2] * (double)ca;
    xc = g.m[3] * (double)cu + g.m[4] * (double)cv + g.m[5] * (double)ca;
  }
  yc += g.dy;
  return make_float2((float)yc, (float)xc);
}

// ------------------------------------------------------------------------------------------
// K_C: Pearson from moments.  Runs as its own tiny kernel, or (single-segment sweeps) as one
// extra workgroup layer of the NEXT batch's k_first_pass, which saves a launch per batch.
// ------------------------------------------------------------------------------------------
struct RefConsts {
  double sw;     // sum of weights = number of masked bins on the full plane
  double swec;   // sum of the float32-rounded w*(E-Ebar) (exactly what K_B multiplies by)
  double var_e;  // sum w (E-Ebar)^2
};

// Pearson coefficient of one candidate from its partial moments: `team` lanes (a whole wavefront,
// or the whole workgroup when it is smaller) stride over the npart slots, the sums are reduced in
// float64 by shuffles, lane 0 of the team returns the score.
__device__ __forceinline__ float pearson_from_moments(double s1, double s2, double s3, const RefConsts& rc) {
  double score = 0.0;
  if (rc.sw > 0) {
    const double var_q = s2 - s1 * s1 / rc.sw;
    const double cov = s3 - (s1 / rc.sw) * rc.swec;
    const double den = var_q * rc.var_e;
    // analysis.py:796-797: zero variance -> 0.  The float32 moments leave O(1e-7) relative
    // rounding in var_q, so "zero" is a relative test.
    if (den > 0 && var_q > 1e-9 * s2) score = cov / sqrt(den);
  }
  return (float)score;
}

__device__ __forceinline__ void team_moments(const double* __restrict__ partials, int npart, int64_t i, int lane,
                                             int team, double& s1, double& s2, double& s3) {
  s1 = s2 = s3 = 0;
  // the triples were written by another kernel a moment ago (HBM / Infinity Cache latency): keep a lane's loads
  // in flight together instead of waiting for each slot in turn
  const double* const base = partials + i * npart * 3;
  int k = lane;
  for (; k + 3 * team < npart; k += 4 * team) {
    double v[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 3; ++e) v[j][e] = base[(size_t)(k + j * team) * 3 + e];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s1 += v[j][0];
      s2 += v[j][1];
      s3 += v[j][2];
    }
  }
  for (; k < npart; k += team) {
    const double* p = base + (size_t)k * 3;
    s1 += p[0];
    s2 += p[1];
    s3 += p[2];
  }
  for (int off = team >> 1; off > 0; off >>= 1) {
    s1 += __shfl_down(s1, off, team);
    s2 += __shfl_down(s2, off, team);
    s3 += __shfl_down(s3, off, team);
  }
}

// scores[i] for i in [0, n): one team per candidate, teams_total teams in flight, this one is team_id
__device__ __forceinline__ void finalize_range(const double* __restrict__ partials, int npart, int n,
                                               const RefConsts& rc, float* __restrict__ scores, int team_id,
                                               int teams_total, int lane, int team) {
  for (int i = team_id; i < n; i += teams_total) {
    double s1, s2, s3;
    team_moments(partials, npart, i, lane, team, s1, s2, s3);
    if (lane == 0) scores[i] = pearson_from_moments(s1, s2, s3, rc);
  }
}

__global__ void k_finalize(const double* __restrict__ partials, int npart, int64_t n, RefConsts rc,
                           float* __restrict__ scores) {
  const int team = blockDim.x < 64 ? blockDim.x : 64;
  const int per_block = blockDim.x / team;
  finalize_range(partials, npart, (int)n, rc, scores, blockIdx.x * per_block + threadIdx.x / team,
                 gridDim.x * per_block, threadIdx.x % team, team);
}

// sums of the partial moments per candidate ([n][3]) for the several-segment scorer
__global__ void k_sum_partials(const double* __restrict__ partials, int npart, int n, double* __restrict__ sums) {
  const int team = 64, per_block = blockDim.x / team;
  const int lane = threadIdx.x % team;
  for (int i = blockIdx.x * per_block + threadIdx.x / team; i < n; i += gridDim.x * per_block) {
    double s1, s2, s3;
    team_moments(partials, npart, i, lane, team, s1, s2, s3);
    if (lane == 0) {
      sums[3 * i] = s1;
      sums[3 * i + 1] = s2;
      sums[3 * i + 2] = s3;
    }
  }
}

struct FinArgs {  // scores of the previous batch, folded into the first pass as grid layer 0 (n == 0: none)
  const double* partials;
  float* scores;
  int n, npart;
  RefConsts rc;
};

// the folded layer: every workgroup of grid layer 0 takes part, one team of lanes per candidate
__device__ __forceinline__ void finalize_layer(const FinArgs& f) {
  const int team = blockDim.x < 64 ? blockDim.x : 64;
  const int per_block = blockDim.x / team;
  finalize_range(f.partials, f.npart, f.n, f.rc, f.scores, blockIdx.x * per_block + threadIdx.x / team,
                 gridDim.x * per_block, threadIdx.x % team, team);
}

// ------------------------------------------------------------------------------------------
// K_A: raster (or image load) + column FFT.  A workgroup owns a band of image columns of one
// candidate and walks it in tiles of 16 columns (8 transforms of 2 packed columns each).
// ------------------------------------------------------------------------------------------
struct FirstArgs {
  const double* params;   // [B][4] (raster mode)
  const double* units;    // [n_units][3] (radius A, azimuth rad, axial A)
  const float* images;    // [B][N][N] (image mode)
  const float2* twtab;    // [N]
  float2* inter;          // [B] x line-blocked half spectrum (inter_index)
  float* raster_out;      // optional [B][N][N]
  FinArgs fin;            // previous batch's scores (one extra workgroup layer, blockIdx.y == B)
  unsigned long long kb_mask;  // bit kb set: ky block kb is needed downstream (others are not stored)
  DevGeom g;
};

// Intermediate half spectrum of one candidate, N/2 x N complex64 in a line-blocked layout:
//   H[kb = ky / 8][pair = x / 2][r = ky % 8][x % 2]
// i.e. one 128-byte line holds eight consecutive ky of one pair of image columns.  K_A's
// wavefronts own column pairs and write whole lines; K_B reads a block of eight ky (N/2 lines,
// contiguous) and transposes it through LDS.  Returns the complex-element index of (ky, pair, 0).
template <int N>
__host__ __device__ constexpr size_t inter_index(int ky, int pair) {
  return ((size_t)((ky >> 3) * (N / 2) + pair) * 8 + (ky & 7)) * 2;
}

constexpr int MODE_RASTER = 0, MODE_IMAGE = 1, MODE_RASTER_OUT = 2;  // 2: also store the raster image

template <int N>
struct KA {
  static constexpr int T = N / 8;            // lanes per FFT
  static constexpr int FPW = T > 64 ? HH_KA_FPW_BIG : 8;  // FFTs per workgroup and tile
  static constexpr int COLS = 2 * FPW;       // image columns per tile
  static constexpr int THREADS = FPW * T;    // == N
  static constexpr int NQ = N >= 256 ? (N / HH_KA_BAND > 0 ? N / HH_KA_BAND : 1) : 1;  // workgroups per candidate
  static constexpr int TPW = (N / COLS) / NQ;            // tiles per workgroup
  static constexpr int BAND = TPW * COLS;                // image columns per workgroup
  static constexpr int BUF = N;              // complex slots per FFT exchange buffer
  static constexpr int CL = 256;             // lattice centres held in LDS at a time (refilled in chunks beyond)
  static constexpr size_t LDS_FFT = (size_t)FPW * BUF * sizeof(float2);
  static constexpr size_t LDS_CENT = (size_t)CL * sizeof(float2);
  static constexpr size_t LDS_TW = (size_t)(TwN<N>::lds_total + 1) * sizeof(float2);  // compact twiddle table
  static constexpr size_t LDS = LDS_FFT + LDS_CENT + LDS_TW;
  static_assert(TPW * NQ * COLS == N, "column tiling");
};

// Add to the group's two packed columns every centre of cent[0..count) whose truncated footprint
// reaches them.  The pixels live in the lanes' FFT input registers: lane t owns rows t + m*T
// (m < 8) of both columns as r[2m] / r[2m+1] (= v[m].x / v[m].y), so the raster needs no LDS image, no zero fill and
// no atomics; a centre's 2R+1 rows touch at most two register slots when T = 64.  Centres are
// visited in lattice order, so every pixel's sum has a fixed order.
template <int N>
__device__ __forceinline__ void raster_pair(float (&r)[16], const float2* cent, int count, const DevGeom& g, int xa,
                                            int t, int lane) {
  constexpr int T = N / 8, TL = T < 64 ? T : 64;
  const int gbase = T >= 64 ? 0 : lane - t;
  const unsigned long long gmask = TL == 64 ? ~0ull : ((1ull << TL) - 1ull);
  const float lo = (float)(xa - g.rpx - 1), hi = (float)(xa + 1 + g.rpx + 1);
  const float qxa = (float)(xa - N / 2) * g.apix, qxb = (float)(xa + 1 - N / 2) * g.apix;  // X of utils.py:94-99
  const float rp = (float)g.rpx;
  const float k2 = g.inv_sigma2 * 1.44269504088896341f;
  int cur = -1;  // slot pair currently gathered in acc_* (T >= 64 path)
  float acc_ax = 0.f, acc_ay = 0.f, acc_bx = 0.f, acc_by = 0.f;
  for (int base = 0; base < count; base += TL) {
    // every lane prepares one candidate centre: position, row range of its footprint, and the
    // squared column offsets for this pair; a hit then only broadcasts the prepared values
    bool hit = false;
    float pyc = 0.f, pea = 0.f, peb = 0.f;
    int py0 = 0, py1 = -1;
    const int ci = base + (t & (TL - 1));
    if (ci < count) {
      const float2 p = cent[ci];
      pyc = p.x;
      const float cx = p.y * g.inv_apix + (float)(N / 2);
      const float cy = p.x * g.inv_apix + (float)(N / 2);
      hit = (cx >= lo) && (cx <= hi) && (cy >= -rp - 1.f) && (cy <= (float)N + rp);
      py0 = max(0, (int)ceilf(cy - rp));
      py1 = min(N - 1, (int)floorf(cy + rp));
      const float dxa = qxa - p.y, dxb = qxb - p.y;
      // a column outside the footprint gets an offset that underflows the exponential to 0
      pea = fabsf((float)xa - cx) <= rp ? dxa * dxa * k2 : 3.0e38f;
      peb = fabsf((float)(xa + 1) - cx) <= rp ? dxb * dxb * k2 : 3.0e38f;
    }
    unsigned long long todo = (__ballot(hit) >> gbase) & gmask;
    while (todo) {
      const int k = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      float yc, ea, eb;
      int y0, y1;
      if constexpr (T >= 64) {  // k is wave-uniform: scalar lane reads, no LDS round trip
        yc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pyc), k));
        ea = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pea), k));
        eb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(peb), k));
        y0 = __builtin_amdgcn_readlane(py0, k);
        y1 = __builtin_amdgcn_readlane(py1, k);
      } else {
        yc = __shfl(pyc, gbase + k, 64);
        ea = __shfl(pea, gbase + k, 64);
        eb = __shfl(peb, gbase + k, 64);
        y0 = __shfl(py0, gbase + k, 64);
        y1 = __shfl(py1, gbase + k, 64);
      }
      // Walk the footprint's rows in chunks of at most T rows (one chunk for T = 64, R <= 31):
      // a chunk touches each lane at most once, in register slot m0 or m0 + 1, and m0 is
      // uniform across the group, so the slot is chosen by multiplying with 0/1 weights.
      for (int ys = y0; ys <= y1; ys += T) {
        const int ye = min(y1, ys + T - 1);
        const int m0 = ys / T;
        const int ya = t + m0 * T;
        const bool in_a = ya >= ys;
        const int y = in_a ? ya : ya + T;
        const float dyv = (float)(y - N / 2) * g.apix - yc;
        const float d2 = dyv * dyv;
        const bool rowok = y <= ye;
        // exp(-(dx^2 + dy^2) / sigma^2) = 2^-(dx^2 k2 + dy^2 k2), k2 = log2(e) / sigma^2
        const float px = rowok ? __builtin_amdgcn_exp2f(fmaf(d2, -k2, -ea)) : 0.f;
        const float py = rowok ? __builtin_amdgcn_exp2f(fmaf(d2, -k2, -eb)) : 0.f;
        const float ax = in_a ? px : 0.f, ay = in_a ? py : 0.f;
        const float bx = in_a ? 0.f : px, by = in_a ? 0.f : py;
        if constexpr (T >= 64) {
          // m0 is wave-uniform.  Consecutive lattice centres mostly fall into the same slot pair
          // (the row coordinate moves by a few pixels per subunit at small twist), so the four
          // contributions are gathered in plain registers and only moved into the slot file
          // r[] — through the VGPR index register (s_set_gpr_idx) — when m0 changes.
          const int i0 = __builtin_amdgcn_readfirstlane(m0);
          if (i0 != cur) {
            if (cur >= 0) {
              const int j1 = cur < 7 ? cur + 1 : 7;
              r[2 * cur] += acc_ax;
              r[2 * cur + 1] += acc_ay;
              r[2 * j1] += acc_bx;  // zero whenever slot cur + 1 would be past the image
              r[2 * j1 + 1] += acc_by;
            }
            cur = i0;
            acc_ax = acc_ay = acc_bx = acc_by = 0.f;
          }
          acc_ax += ax;
          acc_ay += ay;
          acc_bx += bx;
          acc_by += by;
        } else {
#pragma unroll
          for (int m = 0; m < 8; ++m) {
            const float sa = (m == m0) ? 1.f : 0.f, sb = (m == m0 + 1) ? 1.f : 0.f;
            r[2 * m] = fmaf(sa, ax, fmaf(sb, bx, r[2 * m]));
            r[2 * m + 1] = fmaf(sa, ay, fmaf(sb, by, r[2 * m + 1]));
          }
        }
      }
    }
  }
  if constexpr (T >= 64) {
    if (cur >= 0) {
      const int j1 = cur < 7 ? cur + 1 : 7;
      r[2 * cur] += acc_ax;
      r[2 * cur + 1] += acc_ay;
      r[2 * j1] += acc_bx;
      r[2 * j1 + 1] += acc_by;
    }
  }
}

// Fill the LDS centre list with centres [first, first + count) of the candidate.
__device__ __forceinline__ void fill_centres(float2* cent, const Cand& c, const DevGeom& g, const double* units,
                                          int first, int count, int tid, int nthreads) {
  for (int i = tid; i < count; i += nthreads) cent[i] = centre_position(c, g, units, first + i);
}

template <int N, int MODE, bool RESIDENT>
__device__ __forceinline__ void first_pass_tiles(const FirstArgs& a, const size_t b, const Cand& c, int c_lo, int c_hi,
                                                 const TwLds& tw, float2* bufs, float2* cent) {
  using K = KA<N>;
  constexpr int T = K::T;
  const int tid = threadIdx.x;
  const int f = tid / T, t = tid % T;
  float2* const buf = bufs + f * K::BUF;
  const int band0 = blockIdx.x * K::BAND;
  const DevGeom& g = a.g;
#pragma unroll 1
  for (int tile = 0; tile < K::TPW; ++tile) {
    const int x0 = band0 + tile * K::COLS;
    const int xa = x0 + 2 * f;
    float2 v[8];
    if constexpr (MODE != MODE_IMAGE) {
      float r[16];
#pragma unroll
      for (int m = 0; m < 16; ++m) r[m] = 0.f;
      // one pass over the LDS centre list; if the band has more candidate centres than the list
      // holds they are streamed through it in chunks (workgroup-uniform trip count)
#pragma unroll 1
      for (int cb = c_lo; cb < c_hi; cb += K::CL) {
        const int cnt = min(K::CL, c_hi - cb);
        if constexpr (!RESIDENT) {
          __syncthreads();
          fill_centres(cent, c, g, a.units, cb, cnt, tid, K::THREADS);
          __syncthreads();
        }
        if (!(HH_ABLATE & 1)) raster_pair<N>(r, cent, cnt, g, xa, t, tid & 63);
      }
#pragma unroll
      for (int m = 0; m < 8; ++m) v[m] = make_float2(r[2 * m], r[2 * m + 1]);
      if constexpr (MODE == MODE_RASTER_OUT) {
#pragma unroll
        for (int m = 0; m < 8; ++m)
          *reinterpret_cast<float2*>(a.raster_out + (b * N + (size_t)(t + m * T)) * N + xa) = v[m];
      }
    } else {
      const float* img = a.images + b * (size_t)N * N;
#pragma unroll
      for (int m = 0; m < 8; ++m) v[m] = *reinterpret_cast<const float2*>(img + (size_t)(t + m * T) * N + xa);
    }

    if (!(HH_ABLATE & 2)) fft_lanes<N>(v, tw, t, buf);  // v[m] = Z[t + m*T], Z = DFT_y(col_a + i col_b)

    // Split Z into the two real columns' spectra: A[k] = (Z[k] + conj Z[N-k]) / 2,
    // B[k] = (Z[k] - conj Z[N-k]) / (2i), k < N/2; ky = 0 and ky = N/2 (both real) share row 0.
#pragma unroll
    for (int m = 0; m < 8; ++m) buf[t + m * T] = v[m];
    group_sync<T>();
    float4 ab[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int k = t + m * T;
      const float2 zk = v[m];
      const float2 zm = buf[(N - k) & (N - 1)];
      ab[m] = make_float4(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y), 0.5f * (zk.y + zm.y), -0.5f * (zk.x - zm.x));
    }
    if (t == 0) ab[0] = make_float4(v[0].x, v[4].x, v[0].y, v[4].y);
    // Store the pair's 2 x N/2 values in the line-blocked layout H[ky/8][pair][ky%8][2]: eight
    // consecutive lanes (eight consecutive ky) fill one 128-byte line, so a wavefront writes whole
    // lines by itself and the tile loop needs no workgroup barrier and no staging tile.
    float2* const out = a.inter + b * (size_t)(N / 2) * N;
    const int pair = (x0 >> 1) + f;
#pragma unroll
    for (int m = 0; m < ((HH_ABLATE & 4) ? 1 : 4); ++m) {
      const int k = t + m * T;
      // ky blocks the mask never looks at are not written (K_B skips them as well)
      if ((a.kb_mask >> (k >> 3)) & 1ull) *reinterpret_cast<float4*>(out + inter_index<N>(k, pair)) = ab[m];
    }
    group_sync<T>();  // the mirror reads above are done before the next tile's exchange writes
  }
}

template <int N, int MODE>
__global__ __launch_bounds__(KA<N>::THREADS, (N >= 512 ? HH_KA_WPS : 1)) void k_first_pass(FirstArgs a) {
  using K = KA<N>;
  constexpr int T = K::T;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float2* const bufs = reinterpret_cast<float2*>(smem);
  float2* const cent = reinterpret_cast<float2*>(smem + K::LDS_FFT);
  const int tid = threadIdx.x;
  const int f = tid / T, t = tid % T;
  float2* const buf = bufs + f * K::BUF;
  const int band0 = blockIdx.x * K::BAND;
  const DevGeom& g = a.g;

  if constexpr (MODE == MODE_RASTER) {
    if (a.fin.n > 0 && blockIdx.y == 0) {  // the extra layer: scores of the previous batch
      finalize_layer(a.fin);
      return;
    }
  }
  const size_t b = blockIdx.y - ((MODE == MODE_RASTER && a.fin.n > 0) ? 1 : 0);

  // Lattice centres that can reach this band of columns -> LDS, once per workgroup.  The axial
  // coordinate of centre (i, s, u) is m5 * i * rise + O(slack), so only a window of subunit
  // indices i matters; without that monotonicity (steep tilt/psi) every centre is a candidate.
  Cand c;
  int c_lo = 0, c_hi = 0;
  if constexpr (MODE != MODE_IMAGE) {
    c = decode_candidate(a.params + 4 * b, g);
    c_hi = c.M;
    if (g.fast && c.M > 0) {
      const float step = (float)(g.m[5] * c.rise);
      float i0 = ((float)(band0 - g.rpx - 3 - N / 2) * g.apix - g.slack) / step;
      float i1 = ((float)(band0 + K::BAND + g.rpx + 3 - N / 2) * g.apix + g.slack) / step;
      if (i0 > i1) { const float tmp = i0; i0 = i1; i1 = tmp; }
      i0 = fmaxf(i0, -2.0e9f);
      i1 = fminf(i1, 2.0e9f);
      const int ilo = max(-c.imax, (int)floorf(i0) - 1), ihi = min(c.imax, (int)ceilf(i1) + 1);
      const int per = c.csym * g.n_units;
      c_lo = (ilo + c.imax) * per;
      c_hi = ihi < ilo ? c_lo : (ihi + c.imax + 1) * per;
    }
    if (c_hi - c_lo <= K::CL) fill_centres(cent, c, g, a.units, c_lo, c_hi - c_lo, tid, K::THREADS);
  }
  const bool resident = (c_hi - c_lo) <= K::CL;  // workgroup-uniform

  // Twiddles: one compact LDS copy per workgroup, written by the LAST transform group while the
  // first lanes are busy with the float64 centre list; one barrier publishes both.
  float2* const twl = reinterpret_cast<float2*>(smem + K::LDS_FFT + K::LDS_CENT);
  if (f == K::FPW - 1 && !(HH_ABLATE & 8)) fill_twiddles_lds<N>(twl, t, a.twtab);
  __syncthreads();
  const TwLds tw{twl, t};

  // Two copies of the tile loop: the common one never refills the centre list, so the float64
  // trigonometry of the refill cannot raise its register pressure.
  if (resident)
    first_pass_tiles<N, MODE, true>(a, b, c, c_lo, c_hi, tw, bufs, cent);
#ifndef HH_NO_CHUNKED
  else
    first_pass_tiles<N, MODE, false>(a, b, c, c_lo, c_hi, tw, bufs, cent);
#endif
}

// ------------------------------------------------------------------------------------------
// K_A, shared-twist form.  The footprint of a subunit is a product ex(x) * ey(y) (square
// truncation window, Gaussian), so the column transform of the image is
//     H[ky][x] = sum_c ex_c(x) * G_c[ky],      G_c[ky] = sum_y ey_c(y) W_N^(ky y).
// Without tilt/psi the row coordinate of subunit (i, s, u) depends on (twist, csym, rot) only and
// the axial one on (rise, i, u) only.  A sweep visits long runs of candidates that share
// (twist, csym, rot) — the grid is twist-major (app.py:2319-2403) — so G is tabulated once per run
// (summed over the csym copies, which share the axial coordinate) and every candidate of the run
// builds its intermediate from ~(2R+1)/rise_px table rows per column pair: no raster, no column
// transform.  Output layout and everything downstream are the ones of k_first_pass.
// ------------------------------------------------------------------------------------------
struct TableArgs {
  const double* params;   // [B][4]: the batch's candidates (runs of run_len share twist, csym, rot)
  const double* units;
  const float2* twtab;    // [N] exp(-2 pi i k / N)
  float2* table;          // [runs][cap][N/2]: G rows, slot ky = 0 packs (G[0], G[N/2]) — both real
  const int* run_imax;    // [runs] subunit index range [-imax, imax] a run's table covers
  float2* inter;
  FinArgs fin;
  unsigned long long kb_mask;
  int run_len;            // candidates per run inside this batch (>= batch: one run)
  int cap;                // table rows reserved per run
  int rows_lds;           // table rows a workgroup of k_first_pass_table can stage
  DevGeom g;
};

template <int N>
struct KT {
  static constexpr int NKY = N / 2;
  static constexpr int KYW = NKY < HH_KT_KYW ? NKY : HH_KT_KYW;   // ky rows per workgroup
  static constexpr int LANES = KYW < 64 ? KYW : 64;   // lanes of a wavefront that own ky rows
  static constexpr int KPL = KYW / LANES;             // ky rows per lane: ky = ky0 + lane + 64 m
  static constexpr int THREADS = 256, WAVES = 4;
  static constexpr int PAIRS_WG = NKY < HH_KT_PAIRS ? NKY : HH_KT_PAIRS;  // column pairs per workgroup
  static constexpr int COLS = 2 * PAIRS_WG;
  static constexpr int PPW = PAIRS_WG / WAVES;          // ... and per wavefront, taken two at a time
  static constexpr int NQX = NKY / PAIRS_WG, NQY = NKY / KYW;
  static constexpr int ROWS_MAX = 64;                   // table rows a band may need (one lane per row)
  static constexpr int SUB = 4;                         // table rows per workgroup of k_run_table
  static_assert(PPW % 2 == 0, "pairs are processed two at a time");
};

// One table per run: row (i, u) = sum over the csym copies s of G_(i,s,u), for i in [-imax, imax].
// A workgroup takes SUB table rows; the float64 trigonometry of their csym copies' row coordinates
// is done once (one lane per copy, through LDS), then thread ky sums the footprint's rows.
template <int N>
__global__ __launch_bounds__(256) void k_run_table(TableArgs a) {
  using K = KT<N>;
  constexpr int PAIRS = 256;  // (table row, csym copy) pairs resolved per round
  __shared__ float yc_s[PAIRS];
  __shared__ float2 tw_s[N];  // the phase table, gathered at (ky y) mod N: LDS latency instead of L2's
  for (int i = threadIdx.x; i < N; i += 256) tw_s[i] = a.twtab[i];
  const int run = blockIdx.y;
  const DevGeom& g = a.g;
  Cand c = decode_candidate(a.params + 4 * (size_t)run * a.run_len, g);
  c.imax = a.run_imax[run];
  const int rows = (2 * c.imax + 1) * g.n_units;
  float2* const tab = a.table + (size_t)run * a.cap * K::NKY;
  const float rp = (float)g.rpx;
  const float k2 = g.inv_sigma2 * 1.44269504088896341f;
  const int row0 = blockIdx.x * K::SUB;
  const int nrow = min(K::SUB, rows - row0);
  if (nrow <= 0) return;
  float2 acc[K::SUB][(K::NKY + 255) / 256];
  float alt[K::SUB];  // ky = N/2: sum ey (-1)^y (thread 0 only needs it)
#pragma unroll
  for (int r = 0; r < K::SUB; ++r) {
    alt[r] = 0.f;
#pragma unroll
    for (int q = 0; q < (K::NKY + 255) / 256; ++q) acc[r][q] = make_float2(0.f, 0.f);
  }
  const int total = nrow * c.csym;  // pairs, ordered (row, s)
  for (int p0 = 0; p0 < total; p0 += PAIRS) {
    const int np = min(PAIRS, total - p0);
    __syncthreads();
    if ((int)threadIdx.x < np) {
      const int p = p0 + threadIdx.x;
      const int row = row0 + p / c.csym, sc = p % c.csym;
      const int ir = row / g.n_units, u = row % g.n_units;
      yc_s[threadIdx.x] = centre_position(c, g, a.units, (ir * c.csym + sc) * g.n_units + u).x;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < K::SUB; ++r) {
      // pairs of local row r inside this round: [r csym, (r + 1) csym) intersected with [p0, p0 + np)
      const int lo = max(r * c.csym, p0), hi = min((r + 1) * c.csym, p0 + np);
      for (int p = lo; p < hi; ++p) {
        const float yc = yc_s[p - p0];
        const float cy = yc * g.inv_apix + (float)(N / 2);
        if (!(cy >= -rp - 1.f && cy <= (float)N + rp)) continue;
        const int y0 = max(0, (int)ceilf(cy - rp)), y1 = min(N - 1, (int)floorf(cy + rp));
        for (int y = y0; y <= y1; ++y) {
          const float dyv = (float)(y - N / 2) * g.apix - yc;
          const float e = __builtin_amdgcn_exp2f(-dyv * dyv * k2);
          alt[r] += (y & 1) ? -e : e;
#pragma unroll
          for (int q = 0; q < (K::NKY + 255) / 256; ++q) {
            const int ky = threadIdx.x + 256 * q;
            const float2 w = tw_s[(ky * y) & (N - 1)];
            acc[r][q].x = fmaf(e, w.x, acc[r][q].x);
            acc[r][q].y = fmaf(e, w.y, acc[r][q].y);
          }
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < K::SUB; ++r) {
    if (r >= nrow) break;
#pragma unroll
    for (int q = 0; q < (K::NKY + 255) / 256; ++q) {
      const int ky = threadIdx.x + 256 * q;
      if (ky >= K::NKY) continue;
      float2 o = acc[r][q];
      if (ky == 0) o.y = alt[r];
      tab[(size_t)(row0 + r) * K::NKY + ky] = o;
    }
  }
}

// A workgroup owns COLS image columns x KYW ky rows of one candidate and stages in LDS the table
// rows whose subunits can reach its band (at most rows_lds, which the host sized from the smallest
// rise of the sweep).  A wavefront then takes its column pairs two at a time: lane j evaluates the
// four column factors ex(x) of table row j (all rows in parallel), and the rows that reach the four
// columns are accumulated with the factors broadcast by v_readlane (scalar operands) — per row two
// LDS reads of G feed sixteen FMAs.  Each group of two pairs is stored as soon as it is complete.
template <int N>
__global__ __launch_bounds__(256) void k_first_pass_table(TableArgs a) {
  using K = KT<N>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float2* const lds_g = reinterpret_cast<float2*>(smem);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (a.fin.n > 0 && blockIdx.y == 0) {  // the extra layer: scores of the previous batch
    finalize_layer(a.fin);
    return;
  }
  const int band = blockIdx.x % K::NQX, slice = blockIdx.x / K::NQX;
  const int x0 = band * K::COLS, ky0 = slice * K::KYW;
  {
    // ky blocks the mask never looks at are not produced (K_B skips them as well)
    const unsigned long long bits = K::KYW >= 512 ? ~0ull : ((1ull << (K::KYW / 8)) - 1ull) << (ky0 / 8);
    if (!(a.kb_mask & bits)) return;
  }
  const size_t b = blockIdx.y - (a.fin.n > 0 ? 1 : 0);
  const DevGeom& g = a.g;
  const Cand c = decode_candidate(a.params + 4 * b, g);
  const int run = (int)(b / (size_t)a.run_len);
  const int imax_t = a.run_imax[run];
  const float2* const tab = a.table + (size_t)run * a.cap * K::NKY;
  float2* const out = a.inter + b * (size_t)(N / 2) * N;
  const float rp = (float)g.rpx;
  const float k2 = g.inv_sigma2 * 1.44269504088896341f;
  const bool owner = lane < K::LANES;

  // subunit indices whose footprint can reach the band (axial coordinate = z_u + i * rise; the
  // slack covers |z_u| and the float32 roundings of the products)
  int ilo = 0, rows = 0;
  if (c.M > 0) {
    const float inv_rise = 1.0f / (float)c.rise;
    const float i0 = ((float)(x0 - N / 2) * g.apix - rp * g.apix - g.slack) * inv_rise;
    const float i1 = ((float)(x0 + K::COLS - 1 - N / 2) * g.apix + rp * g.apix + g.slack) * inv_rise;
    ilo = max(-c.imax, (int)floorf(fmaxf(i0, -2.0e9f)));
    const int ihi = min(c.imax, (int)ceilf(fminf(i1, 2.0e9f)));
    rows = ihi < ilo ? 0 : min(a.rows_lds, (ihi - ilo + 1) * g.n_units);
  }
  if (!(HH_ABLATE & 1024)) {
    // stage the rows: all of a thread's loads are issued before the first LDS write waits for one
    const float2* const src = tab + ((size_t)(ilo + imax_t) * g.n_units) * K::NKY + ky0;
    constexpr int PER = K::KYW / 2;                 // float4 per staged row
    constexpr int STEP = K::THREADS / PER;          // rows covered by one pass of the workgroup
    const int q = tid % PER, j0 = tid / PER;
#pragma unroll 1
    for (int jb = 0; jb < rows; jb += 8 * STEP) {
      float4 v[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int j = min(jb + j0 + s * STEP, rows - 1);  // unconditional: the loads stay in flight together
        v[s] = *reinterpret_cast<const float4*>(src + (size_t)j * K::NKY + 2 * q);
      }
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int j = min(jb + j0 + s * STEP, rows - 1);  // past the end: the last row again, same data
        *reinterpret_cast<float4*>(lds_g + j * K::KYW + 2 * q) = v[s];
      }
    }
  }
  // lane j: axial position of table row j
  float xc = 0.f, cx = -1.0e9f;
  if (lane < rows) {
    int i = ilo + lane, u = 0;
    if (g.n_units > 1) {
      i = ilo + lane / g.n_units;
      u = lane % g.n_units;
    }
    xc = (float)a.units[3 * u + 2] + (float)((double)i * c.rise);  // utils.py:160, float32 like the lattice list
    cx = xc * g.inv_apix + (float)(N / 2);
  }
  __syncthreads();

#pragma unroll 1
  for (int pg = 0; pg < K::PPW / 2; ++pg) {
    const int pair = (x0 >> 1) + wave * K::PPW + 2 * pg;
    const int xa = 2 * pair;
    float w[4];
    bool any = false;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float dx = (float)(xa + q - N / 2) * g.apix - xc;
      const bool in = fabsf((float)(xa + q) - cx) <= rp;
      w[q] = in ? __builtin_amdgcn_exp2f(-dx * dx * k2) : 0.f;
      any |= in;
    }
    unsigned long long todo = (HH_ABLATE & 256) ? 0ull : __ballot(any);
    float2 acc[4][K::KPL];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int m = 0; m < K::KPL; ++m) acc[q][m] = make_float2(0.f, 0.f);
    while (todo) {  // ascending rows: every sum has a fixed order
      const int j = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      float e[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) e[q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w[q]), j));
      if (owner) {
#pragma unroll
        for (int m = 0; m < K::KPL; ++m) {
          const float2 gk = lds_g[j * K::KYW + lane + 64 * m];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            acc[q][m].x = fmaf(e[q], gk.x, acc[q][m].x);
            acc[q][m].y = fmaf(e[q], gk.y, acc[q][m].y);
          }
        }
      }
    }
    if (owner) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int m = 0; m < K::KPL; ++m) {
          const int k = ky0 + lane + 64 * m;
          if (((a.kb_mask >> (k >> 3)) & 1ull) && (!(HH_ABLATE & 128) || acc[0][m].x == 12345.678f))
            *reinterpret_cast<float4*>(out + inter_index<N>(k, pair + h)) =
                make_float4(acc[2 * h][m].x, acc[2 * h][m].y, acc[2 * h + 1][m].x, acc[2 * h + 1][m].y);
        }
    }
  }
}

// ------------------------------------------------------------------------------------------
// K_B: row FFT + amplitude + masked moments (or: store the half-plane spectrum)
// ------------------------------------------------------------------------------------------
struct SecondArgs {
  const float2* inter;    // [B] x line-blocked half spectrum (inter_index)
  const float2* twtab;    // [N]
  const float2* w2;       // [N/2+1][T][8] {w, w*(E-Ebar)} at kx = t + 64 m: a lane's 8 bins are contiguous (EPI_SCORE)
  double* partials;       // [B][NPART][3]: moments per spectrum row (and wavefront of the row)  (EPI_SCORE)
  float2* spec_out;       // [B][N/2+1][N]                     (EPI_STORE)
  float* q_out;           // EPI_QSTORE: the masked q of every candidate.  Compact (q_roff != NULL, N <= 512): only the bins with
                          // weight, [B][q_stride], row r's bins from q_roff[r] on in ascending kx; else [B][N/2+1][N], rows
                          // lane-major (bin kx = wl + m N/8 at [wl][m])
  const int* q_roff;      // [N/2+1] first compact position of a spectrum row's bins (NULL: the full layout)
  size_t q_stride;        // floats per candidate in q_out
  const int* kb_list;     // ky blocks to process (ascending); NULL = all N/16 of them
  int n_kb;               // entries of kb_list (or N/16)
  int batch;              // candidates in this launch
  int log_flag;
};

constexpr int EPI_SCORE = 0, EPI_STORE = 1, EPI_QSTORE = 2;  // 2: several segments — keep q for the contraction

// Several segments, compact q (round 4): a spectrum row stores only its bins with weight, in ascending kx.  Lane t of the
// row's transform holds the bins kx = t + m T: for a fixed m the row's T lanes are T consecutive bins, so a bin's position
// is the row's offset + the weights counted over the earlier m + the lanes below t with weight in this m — a ballot and
// two population counts per m, once per workgroup (the weights do not change with the candidate).  pos[m] = -1: no weight.
// The same as masks and running bases — for T = 64 both are wave-uniform (scalar registers: the fused pass has no vector
// register to spare) — and the position of lane t's bin m from them: two mbcnt instructions.
template <int T>
struct CompactRow {
  unsigned long long mask[8];
  int base[8];
};
template <int T>
__device__ __forceinline__ CompactRow<T> compact_row(const float (&wx)[8], int base, int lane_in_wave) {
  static_assert(T <= 64, "a row's lanes sit in one wavefront");
  CompactRow<T> r;
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const unsigned long long bal = __ballot(wx[m] > 0.f);
    r.mask[m] = T >= 64 ? bal : (bal >> (lane_in_wave / T * T)) & ((1ull << (T & 63)) - 1ull);
    r.base[m] = base;
    base += __popcll(r.mask[m]);
  }
  return r;
}
template <int T>
__device__ __forceinline__ int compact_pos(const CompactRow<T>& r, int m, int t) {   // -1: the bin has no weight
  return ((r.mask[m] >> t) & 1ull) ? r.base[m] + __popcll(r.mask[m] & ((1ull << t) - 1ull)) : -1;
}
template <int T>
__device__ __forceinline__ void compact_positions(const float (&wx)[8], int base, int t, int lane_in_wave, int pos[8]) {
  static_assert(T <= 64, "a row's lanes sit in one wavefront");
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const unsigned long long bal = __ballot(wx[m] > 0.f);
    const unsigned long long mine = T >= 64 ? bal : (bal >> (lane_in_wave / T * T)) & ((1ull << (T & 63)) - 1ull);
    pos[m] = wx[m] > 0.f ? base + __popcll(mine & ((1ull << t) - 1ull)) : -1;
    base += __popcll(mine);
  }
}

template <int N>
struct KB {
  static constexpr int T = N / 8;             // lanes per FFT
  static constexpr int GROUPS = 8;            // one group per row of a ky block
  static constexpr int THREADS = GROUPS * T;  // == N
  static constexpr int ROWS = N / 2;
  static constexpr int NKB = ROWS / 8;        // ky blocks per candidate
  static constexpr int CPW = HH_KB_CPW;       // candidates per workgroup (one ky block of each)
  static constexpr int WPR = T > 64 ? T / 64 : 1;   // wavefronts per spectrum row
  static constexpr int NPART = (ROWS + 1) * WPR;    // partial-moment slots per candidate: [ky 0..N/2][wavefront]
  static constexpr int BUF = N;               // complex slots per FFT exchange buffer
  static constexpr int PROW = N + 2;          // complex slots per panel row (+16 B: conflict-free b128 writes)
  static constexpr size_t LDS_PANEL = (size_t)8 * PROW * sizeof(float2);
  static constexpr size_t LDS_FFT = (size_t)GROUPS * BUF * sizeof(float2);
  static constexpr size_t LDS = LDS_PANEL + LDS_FFT + (HH_KB_TWLDS ? (size_t)(TwN<N>::lds_total + 1) * sizeof(float2) : 0);
  static constexpr int WAVES_PER_SIMD = HH_KB_WPS;  // register budget (512 / WPS VGPRs)
};

// q = log1p(|F|) or |F| (transforms.py:807-810), up to a constant factor: the Pearson coefficient is
// invariant to scaling q, so the logarithm is taken in base 2 (one v_log_f32, no ln 2 multiply).
// v_sqrt_f32 / v_log_f32 are 1-ulp hardware ops; log2(1 + a) has an ABSOLUTE error of ~1e-7 for
// every a >= 0, which is what the masked sums see (q enters them linearly and squared).
template <int LOG>
__device__ __forceinline__ float amp_to_q(float2 f) {
  const float a = __builtin_amdgcn_sqrtf(f.x * f.x + f.y * f.y);
  if constexpr (LOG) return __log2f(1.0f + a);
  return a;
}

// Sum over the TL = min(T, 64) lanes of a transform group inside one wavefront.  A full wavefront
// uses DPP only (quad permutes and mirrors inside the 16-lane rows, then row_bcast15 / row_bcast31
// across them): no LDS traffic, the total lands in lane 63.  Sub-wavefront groups shuffle and leave
// the total in the group's lane 0.  group_sum_lane<TL>() names that lane.
template <int TL>
__device__ __forceinline__ constexpr int group_sum_lane() { return TL == 64 ? 63 : 0; }

template <int TL>
__device__ __forceinline__ float group_sum(float v) {
  if constexpr (TL == 64) {
#define HH_DPP_ADD(CTRL, ROWS) \
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROWS, 0xF, false))
    HH_DPP_ADD(0xB1, 0xF);   // quad_perm [1,0,3,2]
    HH_DPP_ADD(0x4E, 0xF);   // quad_perm [2,3,0,1]
    HH_DPP_ADD(0x141, 0xF);  // row_half_mirror
    HH_DPP_ADD(0x140, 0xF);  // row_mirror: every lane of a row holds the row's sum
    HH_DPP_ADD(0x142, 0xA);  // row_bcast15 into rows 1 and 3: R0+R1, R2+R3
    HH_DPP_ADD(0x143, 0xC);  // row_bcast31 into rows 2 and 3: lane 63 = R0+R1+R2+R3
#undef HH_DPP_ADD
    return v;
  } else {
#pragma unroll
    for (int off = TL / 2; off > 0; off >>= 1) v += __shfl_down(v, off, TL);
    return v;
  }
}

// Three sums at once.  For a full wavefront the two cross-row steps are written out as v_add_f32_dpp with a row mask
// (one instruction each: the compiler's own lowering of the masked update is a zero fill, a v_mov_b32_dpp and an
// add), interleaved so that no value is read by a DPP instruction less than two instructions after it was written.
template <int TL>
__device__ __forceinline__ void group_sum3(float& a, float& b, float& c) {
  if constexpr (TL == 64) {
#define HH_DPP_ADD3(CTRL)                                                                               \
  a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), CTRL, 0xF, 0xF, false));        \
  b += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(b), CTRL, 0xF, 0xF, false));        \
  c += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c), CTRL, 0xF, 0xF, false))
    HH_DPP_ADD3(0xB1);   // quad_perm [1,0,3,2]
    HH_DPP_ADD3(0x4E);   // quad_perm [2,3,0,1]
    HH_DPP_ADD3(0x141);  // row_half_mirror
    HH_DPP_ADD3(0x140);  // row_mirror: every lane of a row holds the row's sum
#undef HH_DPP_ADD3
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"   // rows 1, 3: R0+R1, R2+R3
        "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"   // rows 2, 3: lane 63 = R0+R1+R2+R3
        "v_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_bcast:31 row_mask:0xc bank_mask:0xf"
        : "+v"(a), "+v"(b), "+v"(c));
  } else {
    a = group_sum<TL>(a);
    b = group_sum<TL>(b);
    c = group_sum<TL>(c);
  }
}

// A workgroup owns ONE ky block (8 spectrum rows, one per transform group) and walks CPW candidates:
// the rows' mask weights and centred reference values are loaded once into registers and serve all
// of them, so the second pass reads the weight table once per 16 candidates instead of once per
// candidate (that re-read cost as much L2 -> CU traffic as the intermediate itself).  Each row's
// three moments are reduced inside its wavefront and written as one partial triple.
template <int N, int EPI, int LOG>
__global__ __launch_bounds__(KB<N>::THREADS, (N >= 256 ? KB<N>::WAVES_PER_SIMD : 1)) void k_second_pass(SecondArgs a) {
  using K = KB<N>;
  constexpr int T = K::T, TL = T < 64 ? T : 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float2* const panel = reinterpret_cast<float2*>(smem);
  float2* const bufs = reinterpret_cast<float2*>(smem + K::LDS_PANEL);
  const int tid = threadIdx.x;
  const int gi = tid / T, t = tid % T;  // group gi owns row gi of the ky block
  float2* const buf = bufs + gi * K::BUF;
  const int kb = a.kb_list ? a.kb_list[blockIdx.x] : (int)blockIdx.x;
  const int row = kb * 8 + gi;
  const int c0 = blockIdx.y * K::CPW;
  const int nc = min(K::CPW, a.batch - c0);
  const size_t cand_stride = (size_t)K::ROWS * N;
  const float2* const in0 = a.inter + (size_t)c0 * cand_stride + (size_t)kb * 8 * N;

#if HH_KB_TWLDS
  float2* const twl = reinterpret_cast<float2*>(smem + K::LDS_PANEL + K::LDS_FFT);
  if (gi == 0) fill_twiddles_lds<N>(twl, t, a.twtab);
  __syncthreads();
  const TwLds twsrc{twl, t};
#else
  float2 tw[TwN<N>::total];
  load_twiddles<N>(tw, t, a.twtab);
  const TwRegs twsrc{tw};
#endif

  // this row's weights {w, w (E - Ebar)} for the lane's 8 bins; the packed row 0 carries ky = 0 and
  // ky = N/2, so its group also keeps the weights of row N/2
  float2 w[8], wn[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) w[m] = wn[m] = make_float2((HH_ABLATE & 64) ? 1.f : 0.f, (HH_ABLATE & 64) ? 0.5f : 0.f);
  if constexpr (EPI != EPI_STORE) {
    if (!(HH_ABLATE & 64)) {
      const float4* const wrow = reinterpret_cast<const float4*>(a.w2 + ((size_t)row * T + t) * 8);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const float4 q = wrow[m];
        w[2 * m] = make_float2(q.x, q.y);
        w[2 * m + 1] = make_float2(q.z, q.w);
      }
      if (row == 0) {
        const float4* const nrow = reinterpret_cast<const float4*>(a.w2 + ((size_t)(N / 2) * T + t) * 8);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const float4 q = nrow[m];
          wn[2 * m] = make_float2(q.x, q.y);
          wn[2 * m + 1] = make_float2(q.z, q.w);
        }
      }
    }
  }
  // several segments, compact q: where this lane's bins go inside a candidate's q (row `row`, and row N/2 for the packed row's group)
  int qpos[8], qposn[8];
  bool compact = false;
  if constexpr (EPI == EPI_QSTORE && T <= 64) {
    compact = a.q_roff != nullptr;
    if (compact) {
      float wx[8], wnx[8];
#pragma unroll
      for (int m = 0; m < 8; ++m) { wx[m] = w[m].x; wnx[m] = wn[m].x; }
      compact_positions<T>(wx, a.q_roff[row], t, tid & 63, qpos);
      compact_positions<T>(wnx, a.q_roff[N / 2], t, tid & 63, qposn);
    }
  }

  // A ky block is N/2 lines = 4N 16-byte pieces, contiguous in memory: piece q = pair*8 + r holds
  // H[ky = 8 kb + r][x = 2 pair, 2 pair + 1].  Every thread moves 4 pieces (coalesced 16 B/lane).
  // (four named registers, not an array: hipcc puts a conditionally re-loaded float4 array in scratch)
  float4 ld0, ld1, ld2, ld3;
  {
    const float4* src = reinterpret_cast<const float4*>(in0) + tid;
    ld0 = src[0];
    ld1 = src[K::THREADS];
    ld2 = src[2 * K::THREADS];
    ld3 = src[3 * K::THREADS];
  }
  // piece q = i * THREADS + tid -> panel slot of (r = q % 8, pair = q / 8)
  auto slot = [&](int i) { const int q = i * K::THREADS + tid; return (q & 7) * K::PROW + 2 * (q >> 3); };
  // one partial triple per (row, wavefront of the row); lane 0 of the group's wavefront writes it
  const int wave_in_row = T > 64 ? (t >> 6) : 0;
  const bool writer = (t & (TL - 1)) == group_sum_lane<TL>();

#pragma unroll 1
  for (int cc = 0; cc < nc; ++cc) {
    const size_t b = (size_t)(c0 + cc);
    // transpose through LDS: panel[r][x]
    *reinterpret_cast<float4*>(panel + slot(0)) = ld0;
    *reinterpret_cast<float4*>(panel + slot(1)) = ld1;
    *reinterpret_cast<float4*>(panel + slot(2)) = ld2;
    *reinterpret_cast<float4*>(panel + slot(3)) = ld3;
    __syncthreads();
    float2 v[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) v[m] = panel[gi * K::PROW + t + m * T];
    __syncthreads();  // the panel may be overwritten; the next candidate's loads fly under this FFT
    if (cc + 1 < nc) {
      const float4* src = reinterpret_cast<const float4*>(in0 + (size_t)(cc + 1) * cand_stride) + tid;
      ld0 = src[0];
      ld1 = src[K::THREADS];
      ld2 = src[2 * K::THREADS];
      ld3 = src[3 * K::THREADS];
    }
    if (!(HH_ABLATE & 16)) fft_lanes<N, HH_FFT_SWZ != 0>(v, twsrc, t, buf);  // v[m] = C[kx = t + m*T]

    if (kb == 0 && (gi == 0 || T > 64)) {
      // Row 0 of H packs two real sequences: C = DFT(F1[0,:]) + i DFT(F1[N/2,:]); un-pack it into the
      // ky = 0 and ky = N/2 rows (all groups of a T > 64 workgroup take the barriers of the exchange).
#pragma unroll
      for (int m = 0; m < 8; ++m) buf[t + m * T] = v[m];  // every group into its own exchange buffer
      group_sync<T>();
      if (gi == 0) {
        float a1 = 0.f, a2 = 0.f, a3 = 0.f, n1 = 0.f, n2 = 0.f, n3 = 0.f;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          const int kx = t + m * T;
          const float2 ck = v[m];
          const float2 cm = buf[(N - kx) & (N - 1)];
          const float2 f0 = make_float2(0.5f * (ck.x + cm.x), 0.5f * (ck.y - cm.y));
          const float2 fn = make_float2(0.5f * (ck.y + cm.y), -0.5f * (ck.x - cm.x));
          if constexpr (EPI != EPI_STORE) {
            const float q0 = amp_to_q<LOG>(f0), qn = amp_to_q<LOG>(fn);
            a1 += w[m].x * q0;
            a2 += w[m].x * q0 * q0;
            n1 += wn[m].x * qn;
            n2 += wn[m].x * qn * qn;
            if constexpr (EPI == EPI_QSTORE) {
              // (q rows are stored lane-major, bin kx = t + m T at [t][m] like W2: a lane's eight bins are 32 contiguous
              // bytes; k_segment_corr only needs q and the segments' spectra in the SAME order)
              if (compact) {
                float* const qb = a.q_out + b * a.q_stride;
                if (qpos[m] >= 0) qb[qpos[m]] = q0;
                if (qposn[m] >= 0) qb[qposn[m]] = qn;
              } else {
                float* const q0row = a.q_out + b * (size_t)(N / 2 + 1) * N;
                q0row[t * 8 + m] = w[m].x > 0.f ? q0 : 0.f;
                q0row[(size_t)(N / 2) * N + t * 8 + m] = wn[m].x > 0.f ? qn : 0.f;
              }
            } else {
              a3 += w[m].y * q0;
              n3 += wn[m].y * qn;
            }
          } else {
            float2* const so = a.spec_out + b * (size_t)(N / 2 + 1) * N;
            so[kx] = f0;
            so[(size_t)(N / 2) * N + kx] = fn;
          }
        }
        if constexpr (EPI != EPI_STORE) {
          group_sum3<TL>(a1, a2, a3);
          group_sum3<TL>(n1, n2, n3);
          if (writer) {
            double* const o = a.partials + (b * K::NPART + wave_in_row) * 3;
            o[0] = a1;
            o[1] = a2;
            o[2] = a3;
            double* const on = a.partials + (b * K::NPART + (size_t)K::ROWS * K::WPR + wave_in_row) * 3;
            on[0] = n1;
            on[1] = n2;
            on[2] = n3;
          }
        }
      }
      group_sync<T>();  // the mirror reads are done before the next transform's exchange writes
      if (gi == 0) continue;
    }

    if constexpr (EPI != EPI_STORE) {
      float s1 = 0.f, s2 = 0.f, s3 = 0.f;
      float* const qrow = (EPI == EPI_QSTORE) ? a.q_out + (b * (size_t)(N / 2 + 1) + row) * N : nullptr;
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const float q = (HH_ABLATE & 32) ? v[m].x + v[m].y : amp_to_q<LOG>(v[m]);
        s1 += w[m].x * q;
        s2 += w[m].x * q * q;
        if constexpr (EPI == EPI_QSTORE) {
          if (compact) {
            if (qpos[m] >= 0) a.q_out[b * a.q_stride + qpos[m]] = q;
          } else {
            qrow[t * 8 + m] = w[m].x > 0.f ? q : 0.f;  // bins outside the mask carry no weight in any segment
          }
        } else {
          s3 += w[m].y * q;
        }
      }
      group_sum3<TL>(s1, s2, s3);
      if (writer) {
        double* const o = a.partials + (b * K::NPART + (size_t)row * K::WPR + wave_in_row) * 3;
        o[0] = s1;
        o[1] = s2;
        o[2] = s3;
      }
    } else {
      float2* const so = a.spec_out + (b * (size_t)(N / 2 + 1) + row) * N;
#pragma unroll
      for (int m = 0; m < 8; ++m) so[t + m * T] = v[m];
    }
  }
}

// ------------------------------------------------------------------------------------------
// Fused pass for shared-twist runs: the intermediate never leaves the compute unit.
//   H[ky][x] = sum_c ex_c(x) G_c[ky]   (k_first_pass_table's identity)
// needs, for one ky block (8 rows), only the 8-ky slice of the run's table (rows x 64 B, a few KB)
// and the candidate's column factors ex_c(x).  A workgroup therefore owns one ky block of up to
// CPW candidates of one run: it keeps the table slice in LDS (transposed, [ky][row]), builds each
// candidate's 8 x N panel of H straight into the row transforms' exchange buffers, and runs
// k_second_pass's transform + moments on it.  HBM sees the parameters, the column factors
// (k_column_factors: KG x N floats per candidate, written once, read from L2) and the scores.
// ------------------------------------------------------------------------------------------
struct FactorArgs {
  const double* params;   // [B][4]
  const double* units;
  const int* run_imax;    // [runs]
  float* eg;              // [B][kg][N] column factors: eg[k][x] = ex of table row cg(x/4) + k at column x
  int* cgs;               // [B][N/4 + 4] first table row of every group of four columns, then the candidate's row count
  int run_len, kg, rows_lds;
  int count;              // candidates (0: nothing to do)
  DevGeom g;
};

// ints per candidate in the cgs array: the N/4 groups' first table rows, then (one 16-byte piece) the number of table
// rows the candidate's build has to walk
template <int N>
__host__ __device__ constexpr int cgs_stride() { return N / 4 + 4; }

// lds: workgroup scratch of at least (rows_lds + 1) floats — the axial coordinate of every table row of the candidate
// (it depends on the row only, so its float64 product is done once per row, not once per row and column) and the
// workgroup's row-count maximum.
template <int N>
__device__ __forceinline__ void column_factors_of(const FactorArgs& a, int b, float* lds) {
  float* const xc_tab = lds;                                   // [rows_lds]
  int* const kc_s = reinterpret_cast<int*>(lds + a.rows_lds);  // max over the groups of (last - first + 1)
  const int x = threadIdx.x;
  const DevGeom& g = a.g;
  const Cand c = decode_candidate(a.params + 4 * (size_t)b, g);
  const int imax_t = a.run_imax[b / a.run_len];
  const int rows = (2 * imax_t + 1) * g.n_units;
  const float rp = (float)g.rpx;
  const float k2 = g.inv_sigma2 * 1.44269504088896341f;
  if (x == 0) *kc_s = 0;
  for (int r = x; r < a.rows_lds; r += N) {
    int i = r - imax_t, u = 0;
    if (g.n_units > 1) {
      i = r / g.n_units - imax_t;
      u = r % g.n_units;
    }
    const bool ok = r < rows && i >= -c.imax && i <= c.imax;
    // utils.py:160, float32 like the lattice list; rows the candidate does not have never pass the window test
    xc_tab[r] = ok ? (float)a.units[3 * u + 2] + (float)((double)i * c.rise) : __builtin_nanf("");
  }
  const int x0 = x & ~3;
  // first subunit index whose footprint can reach the group's columns (axial coordinate z_u + i rise;
  // the slack covers |z_u| and the float32 roundings): a conservative start, kg rows from it cover the group
  int cg = 0;
  if (c.M > 0) {
    const float lo = ((float)(x0 - N / 2) * g.apix - rp * g.apix - g.slack) / (float)c.rise;
    const int i0 = max(-c.imax, (int)ceilf(fmaxf(lo, -2.0e9f)));
    cg = (min(i0, c.imax) + imax_t) * g.n_units;
  }
  cg = max(0, min(cg, a.rows_lds - a.kg));
  __syncthreads();
  // Tighten the window per group: rows [first, last] of the kg candidates that reach ANY of the group's four
  // columns, by the SAME truncation test the raster and the run-table first pass apply.  The group's base moves to
  // cg + first and the fused pass walks only max-over-groups(last - first + 1) rows of this candidate (rows past a
  // group's own count carry zero factors, as before).
  unsigned hit = 0;
  for (int k = 0; k < a.kg; ++k) {
    const float cx = xc_tab[cg + k] * g.inv_apix + (float)(N / 2);
    hit |= fabsf((float)x - cx) <= rp ? (1u << k) : 0u;
  }
  hit |= __shfl_xor((int)hit, 1, 64);
  hit |= __shfl_xor((int)hit, 2, 64);
  const int first = hit ? __ffs((int)hit) - 1 : 0;
  const int count = hit ? (32 - __clz((int)hit)) - first : 0;
  cg += first;
  if ((x & 3) == 0 && count > 0) atomicMax(kc_s, count);
  __syncthreads();
  // only the rows the fused pass will walk are written (at least one: a candidate that reaches no column still gets a
  // row of zeros); what the buffer holds beyond them is never read
  const int kc = *kc_s, kw = min(a.kg, max(kc, 1));
  for (int k = 0; k < kw; ++k) {
    float wgt = 0.f;
    if (k < count) {
      const float xc = xc_tab[cg + k];
      const float cx = xc * g.inv_apix + (float)(N / 2);
      const float dx = (float)(x - N / 2) * g.apix - xc;
      if (fabsf((float)x - cx) <= rp) wgt = __builtin_amdgcn_exp2f(-dx * dx * k2);
    }
    a.eg[((size_t)b * a.kg + k) * N + x] = wgt;
  }
  int* const cgo = a.cgs + (size_t)b * cgs_stride<N>();
  if ((x & 3) == 0) cgo[x >> 2] = cg;
  if (x < 4) cgo[N / 4 + x] = kc;
#if HH_POISON
  // sanitizer build: factor rows past the candidate's row count must never be read by the fused pass
  for (int k = kw; k < a.kg; ++k) a.eg[((size_t)b * a.kg + k) * N + x] = __builtin_nanf("");
#endif
}

template <int N>
__global__ __launch_bounds__(N) void k_column_factors(FactorArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_f[];  // (rows_lds + 1) floats
  column_factors_of<N>(a, blockIdx.x, reinterpret_cast<float*>(smem_f));
}

struct FusedArgs {
  const double* params;
  const float2* twtab;
  const float2* table;    // [runs][cap][N/2]
  const int* run_imax;    // [runs]
  const float* eg;        // [B][kg][N]
  const int* cgs;         // [B][N/4 + 4]
  const float2* w2;
  double* partials;
  float* q_out;           // EPI_QSTORE (layout: SecondArgs)
  const int* q_roff;
  size_t q_stride;
  const int* kb_list;
  int n_kb;
  int batch;              // candidates in this launch
  int run_len;            // candidates per run inside this batch
  // work layers: the first runs_a runs are cut into groups_a layers of cpw_a candidates each, the remaining runs
  // into groups_b layers of cpw_b (shorter workgroups for the launch's last, partly filled round: fused_schedule)
  int runs_a, groups_a, cpw_a, groups_b, cpw_b;
  int cap;                // table rows reserved per run
  int rows_lds;           // table rows staged per ky (>= every run's row count, >= kg)
  int kg;                 // table rows per group of four columns
  int n_units;
  int factor_layers;      // leading grid layers that compute the NEXT batch's column factors (next.count of them)
  FinArgs fin;
  FactorArgs next;
};

typedef __attribute__((address_space(1))) const void* gptr_t;  // operands of __builtin_amdgcn_global_load_lds
typedef __attribute__((address_space(3))) void* lptr_t;

// One LDS-DMA wave-instruction: every active lane moves 16 bytes from its own global address to
// lds_base + 16 * lane (lds_base wave-uniform).  Issued through inline assembly on purpose: for the builtin, the
// compiler's alias tracking of LDS-DMA cannot tell the factor buffer being filled from the one being read and puts a
// conservative s_waitcnt vmcnt(0) in front of the next ds_read, which serialises the copy with the transform it is
// meant to fly under.  Here the compiler does not see the LDS write at all, so the CALLER orders it: s_waitcnt
// vmcnt(0) (lds_dma_wait) and then a workgroup barrier before any wave reads the destination.
// (M0 is a reserved register, which clang warns about on a clobber list; the clobber is still what tells the backend's
// M0-initialisation hoisting that the register does not survive this statement.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void lds_dma16(const void* gsrc_lane, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
               :
               : "v"(gsrc_lane), "s"(lds_base)
               : "memory", "m0");  // M0 is written here: the compiler must not keep a value of its own in it across this
}
#pragma clang diagnostic pop
__device__ __forceinline__ void lds_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned lds_offset_of(const void* p) {  // wave-uniform LDS byte address as a scalar
  return (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(lptr_t)(const_cast<void*>(p)));
}

template <int N>
struct KF {
  static constexpr int T = N / 8;
  static constexpr int THREADS = N;
  static constexpr int BROW = N + 4;               // complex slots per panel row (+32 B against bank conflicts)
  static constexpr size_t LDS_BUF = (size_t)8 * BROW * sizeof(float2);
  static constexpr size_t LDS_R2 = (HH_KF_SPLIT && N == 1024) ? (size_t)(N / 2) * sizeof(float2) : 0;  // W_N^n, n < N/2
  static size_t lds(int rows_lds, int kg) {
    return LDS_BUF + (size_t)8 * rows_lds * sizeof(float2) + 2 * ((size_t)kg * N * sizeof(float) + (size_t)cgs_stride<N>() * sizeof(int)) + LDS_R2;
  }
};

template <int N, int EPI, int LOG>
__global__ __launch_bounds__(N, (N >= 256 ? HH_KF_WPS : 1)) void k_fused_pass(FusedArgs a) {
  using K = KF<N>;
  using KP = KB<N>;  // the partial-moment layout is k_second_pass's
  constexpr int T = K::T, TL = T < 64 ? T : 64, NKY = N / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float2* const bufs = reinterpret_cast<float2*>(smem);
  float2* const gs = reinterpret_cast<float2*>(smem + K::LDS_BUF);                       // [8][rows_lds]
  float* const eg = reinterpret_cast<float*>(smem + K::LDS_BUF + (size_t)8 * a.rows_lds * sizeof(float2));  // [2][kg][N]
  int* const cgs = reinterpret_cast<int*>(eg + (size_t)2 * a.kg * N);                    // [2][N/4 + 4]
  constexpr int CGS = cgs_stride<N>();
  const int tid = threadIdx.x;
  // leading grid layers: the next batch's column factors (so that batch needs no launch of its own),
  // then the scores of the previous batch
  if ((int)blockIdx.y < a.factor_layers) {
    const int cand = blockIdx.y * gridDim.x + blockIdx.x;
    if (cand < a.next.count) column_factors_of<N>(a.next, cand, reinterpret_cast<float*>(smem));  // these layers use no other LDS
    return;
  }
  if (a.fin.n > 0 && (int)blockIdx.y == a.factor_layers) {
    finalize_layer(a.fin);
    return;
  }
  const int gi = tid / T, t = tid % T;
  float2* const buf = bufs + gi * K::BROW;
  // Work layers -> (layer of candidates gy, ky block): workgroups are dealt round-robin over the 8 XCDs in dispatch
  // order, so ids that are equal mod 8 share an L2.  Every ky block of one layer reads the same column factors and
  // neighbouring 64-byte pieces of the same table rows; mapping a layer's blocks to ids of ONE residue class lets the
  // XCD's L2 fetch them once instead of all eight L2s once each (placement is a speed matter only).
  int gy = blockIdx.y - a.factor_layers - (a.fin.n > 0 ? 1 : 0);
  int kbi = blockIdx.x;
#if HH_XCD_MAP
  {
    const int nkb = gridDim.x, work_layers = gridDim.y - a.factor_layers - (a.fin.n > 0 ? 1 : 0);
    const int lw = gy * nkb + kbi;
    if (lw < (work_layers & ~7) * nkb) {
      const int seq = lw >> 3;
      gy = (seq / nkb) * 8 + (lw & 7);
      kbi = seq % nkb;
    }
  }
#endif
  const int kb = a.kb_list ? a.kb_list[kbi] : kbi;
  const int row = kb * 8 + gi;
  int run, off, cpw;
  {
    const int la = a.runs_a * a.groups_a;
    if (gy < la) {
      run = gy / a.groups_a;
      off = (gy % a.groups_a) * a.cpw_a;
      cpw = a.cpw_a;
    } else {
      run = a.runs_a + (gy - la) / a.groups_b;
      off = ((gy - la) % a.groups_b) * a.cpw_b;
      cpw = a.cpw_b;
    }
  }
  const int cfirst = run * a.run_len + off;
  const int nc = min(cpw, min(a.run_len - off, a.batch - cfirst));
  if (nc <= 0) return;

  // N = 1024 (SPLIT): a row belongs to two wavefronts (h = 0, 1).  A lane's build already owns columns n and n + N/2
  // (its two groups of four), so the first radix-2 step of the row transform needs no exchange:
  //   y0[n] = x[n] + x[n + N/2],  y1[n] = (x[n] - x[n + N/2]) W_N^n,   X[2k] = DFT_{N/2}(y0)[k],  X[2k + 1] = DFT_{N/2}(y1)[k].
  // Wavefront h then transforms y_h (N/2 = 512 points, exchanges inside the wavefront, no workgroup barrier) and
  // scores the bins kx = 2k + h.  One workgroup barrier per candidate (both wavefronts write both halves) replaces
  // the six of the 8 x 8 x 8 x 2 plan, whose every exchange crossed the two wavefronts.
  constexpr bool SPLIT = HH_KF_SPLIT && N == 1024;
  constexpr int NF = SPLIT ? N / 2 : N, TF = NF / 8;   // transform length and lanes of one transform
  const int h = SPLIT ? (t >> 6) : 0, tf = SPLIT ? (t & 63) : t;
  float2* const fbuf = SPLIT ? buf + h * (NF + 2) : buf;   // the wavefront's half of the row's panel (16-byte aligned)
  const int wl = SPLIT ? 2 * tf + h : t;   // the lane's slot in a weight row: bins kx = wl + m T  (SPLIT: 2 (tf + 64 m) + h)

  float2 tw[TwN<NF>::total];
  load_twiddles<NF, N / NF>(tw, tf, a.twtab);
  const TwRegs twsrc{tw};
  // SPLIT: the radix-2 step's twiddles W_N^n (n < N/2) live in LDS (a lane reads its four per candidate: registers
  // are what this kernel is short of)
  float2* const r2tab = reinterpret_cast<float2*>(cgs + 2 * CGS);
  if constexpr (SPLIT) {
    for (int e = tid; e < N / 2; e += K::THREADS) r2tab[e] = a.twtab[e];
  }

  float2 w[8];  // this row's weights {w, w (E - Ebar)} for the lane's 8 bins, kept for all candidates
  {
    const float4* const wrow = reinterpret_cast<const float4*>(a.w2 + ((size_t)row * T + wl) * 8);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const float4 q = wrow[m];
      w[2 * m] = make_float2(q.x, q.y);
      w[2 * m + 1] = make_float2(q.z, q.w);
    }
  }

  // several segments, compact q: where this lane's bins go inside a candidate's q
  // (positions are worked out at the store — one compare, two mbcnt, one scalar count per bin: this kernel has neither a
  // vector nor a scalar register left to keep eight masks and bases across the candidates)
  bool compact = false;
  int qrow_base = 0;
  if constexpr (EPI == EPI_QSTORE && T <= 64) {
    compact = a.q_roff != nullptr;
    if (compact) qrow_base = a.q_roff[row];
  }
  // candidate b's eight q of this lane -> its compact row
  auto store_q_compact = [&](size_t b, const float (&qv)[8]) {
    if constexpr (T <= 64) {
      float* const qb = a.q_out + ((HH_ABLATE & 65536) ? (b & 63) : b) * a.q_stride;   // (65536: timing only — every candidate's q into 64 cache-resident rows)
      int base = qrow_base;
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const unsigned long long bal = __ballot(w[m].x > 0.f);
        const unsigned long long mine = T >= 64 ? bal : (bal >> ((tid & 63) / T * T)) & ((1ull << (T & 63)) - 1ull);
        if (w[m].x > 0.f) qb[base + __popcll(mine & ((1ull << (t & 63)) - 1ull))] = qv[m];
        base += __popcll(mine);
      }
    }
  };

  // the run's table slice of this ky block, transposed to [ky in block][table row]; rows past the
  // run's own count are zero
  {
    const int rows = (2 * a.run_imax[run] + 1) * a.n_units;
    const float2* const tab = a.table + (size_t)run * a.cap * NKY + 8 * kb;
    for (int e = tid; e < a.rows_lds * 4; e += K::THREADS) {
      const int cr = e >> 2, part = e & 3;
      const float4 q = *reinterpret_cast<const float4*>(tab + (size_t)min(cr, rows - 1) * NKY + 2 * part);
      const bool in = cr < rows;
      gs[(2 * part) * a.rows_lds + cr] = in ? make_float2(q.x, q.y) : make_float2(0.f, 0.f);
      gs[(2 * part + 1) * a.rows_lds + cr] = in ? make_float2(q.z, q.w) : make_float2(0.f, 0.f);
    }
  }
  // column factors of the first candidate (later ones are fetched under the previous transform)
  const int n_e4 = a.kg * (N / 4);  // float4 pieces of one candidate's factors
  auto stage_factors = [&](int b) {
    const float4* const src = reinterpret_cast<const float4*>(a.eg + (size_t)b * a.kg * N);
    for (int e = tid; e < n_e4; e += K::THREADS) reinterpret_cast<float4*>(eg)[e] = src[e];
    if (tid < CGS) cgs[tid] = a.cgs[(size_t)b * CGS + tid];
  };
  constexpr bool EGLOBAL = HH_KF_EGLOBAL && T == 64 && !(HH_KF_SPLIT && N == 1024);
  if constexpr (!EGLOBAL) stage_factors(cfirst);  // buffer 0
  // Every load issued so far (twiddles, weights, slice, factors) is retired HERE, explicitly: the barrier's fence only
  // waits for LDS traffic, and with register loads still pending at the loop's entry the compiler guards their first
  // uses INSIDE the loop with counted waits — the last of them a vmcnt(0) in the middle of part B, which in every later
  // round drains the LDS-DMA copies that are meant to fly until the round's end.
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt and lgkmcnt untouched
  __syncthreads();

  const int wave_in_row = T > 64 ? (t >> 6) : 0;
  const bool writer = (t & (TL - 1)) == group_sum_lane<TL>();

  // One candidate = part A (build the row from the table slice and the candidate's factors, first butterflies, data
  // left in the group's exchange buffer) + part B (rest of the transform, amplitudes, moments).  A is LDS-heavy, B is
  // vector-heavy.  With HH_KF_STAGGER (N = 512: a transform group is one wavefront, the exchanges need no workgroup
  // barrier) wavefronts 4-7 — the SIMD partners of 0-3 — run half a candidate behind: in round `it` wavefronts 0-3 do
  // A(it) B(it), wavefronts 4-7 do B(it-1) A(it), so a SIMD's two wavefronts of this workgroup are in different parts
  // most of the time instead of hitting the LDS and then the vector pipe together.  Nothing of a candidate lives in
  // registers between A and B, and the factor buffers are used exactly as without the stagger: A(it) reads buffer
  // it & 1 in round it, the copies for it + 1 go to the other buffer, one workgroup barrier closes the round.
  constexpr bool STAGGER = HH_KF_STAGGER && T == 64;
  constexpr int KCUT = (HH_KF_CUT < Plan<NF>::n) ? HH_KF_CUT : 1;  // A ends at the exchange after this stage
  const bool late = STAGGER && __builtin_amdgcn_readfirstlane(tid >> 6) >= (N / 64) / 2;
  if (HH_KF_PRIO && late) __builtin_amdgcn_s_setprio(HH_KF_PRIO);

  // (see flush_q) — not for the packed row's group (its two rows are un-packed through the exchange buffer)
  const bool defer_q = EPI == EPI_QSTORE && STAGGER && HH_KF_DEFER_Q && !late && !(kb == 0 && gi == 0);

  auto part_a = [&](int cc) {
    const int cur = cc & 1;
    const float* egc;
    const int* cgc;
    if constexpr (EGLOBAL) {
      egc = a.eg + (size_t)(cfirst + cc) * a.kg * N;
      cgc = a.cgs + (size_t)(cfirst + cc) * CGS;
    } else {
      egc = eg + (size_t)cur * a.kg * N;
      cgc = cgs + cur * CGS;
    }
    // ---- this group's row of H, built by the group itself into its own exchange buffer (no workgroup barrier).
    // A lane owns two groups of four consecutive columns (x = 4 t + c and 4 (t + T) + c): two independent
    // accumulation chains, and the operands of the next table row are in flight while this row's FMAs issue.
    {
      // table rows this candidate needs (<= kg; a wave-uniform LDS word written by the factor kernel)
      // (clamped to the buffer's kg: whatever the word holds, the walk is bounded)
      // (at least one: a candidate that reaches no column has a first factor row of zeros, so the sums need no
      // separate zero fill)
      const int kgn = (HH_ABLATE & 2048) ? 0 : max(1, min(a.kg, __builtin_amdgcn_readfirstlane(cgc[N / 4])));
      // sums of the two column groups xg0, xg1 (four columns each).  The first table row initialises the sums (no zero
      // fill), the others accumulate; the operand addresses are base + k x constant, so the unrolled loop addresses
      // them with instruction offsets.
      float2 p0, p1, p2, p3, q0, q1, q2, q3;
      auto accumulate = [&](int xg0, int xg1) {
        const float2* const grow0 = gs + gi * a.rows_lds + cgc[xg0];
        const float2* const grow1 = gs + gi * a.rows_lds + cgc[xg1];
        const float* const erow0 = egc + 4 * xg0;
        const float* const erow1 = egc + 4 * xg1;
        if (!(HH_ABLATE & 2048)) {
          {
            const float2 ga = grow0[0], gb = grow1[0];
            const float4 ea = *reinterpret_cast<const float4*>(erow0), eb = *reinterpret_cast<const float4*>(erow1);
            p0 = make_float2(ea.x * ga.x, ea.x * ga.y); p1 = make_float2(ea.y * ga.x, ea.y * ga.y);
            p2 = make_float2(ea.z * ga.x, ea.z * ga.y); p3 = make_float2(ea.w * ga.x, ea.w * ga.y);
            q0 = make_float2(eb.x * gb.x, eb.x * gb.y); q1 = make_float2(eb.y * gb.x, eb.y * gb.y);
            q2 = make_float2(eb.z * gb.x, eb.z * gb.y); q3 = make_float2(eb.w * gb.x, eb.w * gb.y);
          }
#pragma unroll 2
          for (int k = 1; k < kgn; ++k) {
            const float2 ga = grow0[k], gb = grow1[k];
            const float4 ea = *reinterpret_cast<const float4*>(erow0 + (size_t)k * N);
            const float4 eb = *reinterpret_cast<const float4*>(erow1 + (size_t)k * N);
            p0.x = fmaf(ea.x, ga.x, p0.x); p0.y = fmaf(ea.x, ga.y, p0.y);
            p1.x = fmaf(ea.y, ga.x, p1.x); p1.y = fmaf(ea.y, ga.y, p1.y);
            p2.x = fmaf(ea.z, ga.x, p2.x); p2.y = fmaf(ea.z, ga.y, p2.y);
            p3.x = fmaf(ea.w, ga.x, p3.x); p3.y = fmaf(ea.w, ga.y, p3.y);
            q0.x = fmaf(eb.x, gb.x, q0.x); q0.y = fmaf(eb.x, gb.y, q0.y);
            q1.x = fmaf(eb.y, gb.x, q1.x); q1.y = fmaf(eb.y, gb.y, q1.y);
            q2.x = fmaf(eb.z, gb.x, q2.x); q2.y = fmaf(eb.z, gb.y, q2.y);
            q3.x = fmaf(eb.w, gb.x, q3.x); q3.y = fmaf(eb.w, gb.y, q3.y);
          }
        } else {
          p0 = p1 = p2 = p3 = q0 = q1 = q2 = q3 = make_float2(0.f, 0.f);
        }
      };
      // The row is handed to the transform through the group's exchange buffer.  A lane stores 2 x 32 bytes at a
      // 32-byte lane stride: the eight lanes of a ds_write_b128 group would hit four bank groups twice, so the
      // 16-byte chunk c = x / 2 lives at c ^ ((c >> 3) & 1) (chunks 2 xg, 2 xg + 1 of lanes xg and xg + 4 then fall
      // into different halves of the 128-byte bank span); the reader un-swizzles with one precomputed base.
      auto chunk = [](int c) { return HH_KF_PSWZ ? (c ^ ((c >> 3) & 1)) : c; };
      if constexpr (SPLIT) {
        accumulate(t, t + T);
        // the radix-2 step, then y0 into the row's first half and y1 into its second (chunks 2 t, 2 t + 1 of each)
        const float2 d0 = csub(p0, q0), d1 = csub(p1, q1), d2 = csub(p2, q2), d3 = csub(p3, q3);
        p0 = cadd(p0, q0); p1 = cadd(p1, q1); p2 = cadd(p2, q2); p3 = cadd(p3, q3);
        const float4 wa = *reinterpret_cast<const float4*>(r2tab + 4 * t), wb = *reinterpret_cast<const float4*>(r2tab + 4 * t + 2);
        q0 = cmul(d0, make_float2(wa.x, wa.y)); q1 = cmul(d1, make_float2(wa.z, wa.w));
        q2 = cmul(d2, make_float2(wb.x, wb.y)); q3 = cmul(d3, make_float2(wb.z, wb.w));
        float4* const y0 = reinterpret_cast<float4*>(buf);
        float4* const y1 = reinterpret_cast<float4*>(buf + NF + 2);
        y0[chunk(2 * t)] = make_float4(p0.x, p0.y, p1.x, p1.y);
        y0[chunk(2 * t + 1)] = make_float4(p2.x, p2.y, p3.x, p3.y);
        y1[chunk(2 * t)] = make_float4(q0.x, q0.y, q1.x, q1.y);
        y1[chunk(2 * t + 1)] = make_float4(q2.x, q2.y, q3.x, q3.y);
      } else {
        const int xg0 = t, xg1 = t + T;
        accumulate(xg0, xg1);
        float4* const row4 = reinterpret_cast<float4*>(buf);
        row4[chunk(2 * xg0)] = make_float4(p0.x, p0.y, p1.x, p1.y);
        row4[chunk(2 * xg0 + 1)] = make_float4(p2.x, p2.y, p3.x, p3.y);
        row4[chunk(2 * xg1)] = make_float4(q0.x, q0.y, q1.x, q1.y);
        row4[chunk(2 * xg1 + 1)] = make_float4(q2.x, q2.y, q3.x, q3.y);
      }
    }
    // (SPLIT: the one workgroup barrier of the transform.  Tried and dropped: an LDS-counter meeting of just the row's
    // two wavefronts, -3.5 %; every wavefront building its own half's input from all the row's columns — twice the
    // accumulation, no barrier, stagger possible — -16 %)
    group_sync<T>();
    float2 v[8];
    if constexpr (SPLIT) {  // n = tf + 64 m of the wavefront's own half
      const int ps = HH_KF_PSWZ ? ((((tf >> 1) ^ ((tf >> 4) & 1)) << 1) | (tf & 1)) : tf;
#pragma unroll
      for (int m = 0; m < 8; ++m) v[m] = fbuf[ps + m * TF];
    } else if constexpr (HH_KF_PSWZ && T % 32 == 0) {  // x = t + m T: bit 4 of x is bit 4 of t, one swizzled base serves every m
      const int ps = (((t >> 1) ^ ((t >> 4) & 1)) << 1) | (t & 1);
#pragma unroll
      for (int m = 0; m < 8; ++m) v[m] = buf[ps + m * T];
    } else {
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int x = t + m * T;
        v[m] = buf[HH_KF_PSWZ ? ((((x >> 1) ^ ((x >> 4) & 1)) << 1) | (x & 1)) : x];
      }
    }
    if constexpr (T > 64 && !SPLIT) __syncthreads();  // both wavefronts of a row have read it before either exchanges in it
    if (!(HH_ABLATE & 16)) fft_lanes_part<NF, HH_FFT_SWZ != 0, TwRegs, 1, KCUT>(v, twsrc, tf, fbuf);
  };

  auto part_b = [&](int cc) {
    const size_t b = (size_t)(cfirst + cc);
    float2 v[8];
    if (!(HH_ABLATE & 16)) {
      fft_lanes_part<NF, HH_FFT_SWZ != 0, TwRegs, 2, KCUT>(v, twsrc, tf, fbuf);  // v[m] = C[kx = t + m*T] (SPLIT: kx = 2 (tf + 64 m) + h); the exchanges reuse the row's panel slots
    } else {
#pragma unroll
      for (int m = 0; m < 8; ++m) v[m] = fbuf[tf + m * TF];
    }

    bool scored = false;
    if (kb == 0 && (gi == 0 || (T > 64 && !SPLIT))) {
      // Row 0 of H packs two real sequences (see k_second_pass): un-pack into ky = 0 and ky = N/2
#pragma unroll
      for (int m = 0; m < 8; ++m) fbuf[tf + m * TF] = v[m];
      group_sync<TF>();
      if (gi == 0) {
        // the packed row also carries ky = N/2: its weights come from L2 here (one wavefront in 256)
        const float2* const nrow = a.w2 + ((size_t)(N / 2) * T + wl) * 8;
        float a1 = 0.f, a2 = 0.f, a3 = 0.f, n1 = 0.f, n2 = 0.f, n3 = 0.f;
        [[maybe_unused]] int nbase = (EPI == EPI_QSTORE && compact) ? a.q_roff[N / 2] : 0, zbase = qrow_base;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          const float2 wnm = nrow[m];
          const int kx = SPLIT ? 2 * (tf + m * TF) + h : t + m * T;
          const float2 ck = v[m];
          // C[N - kx]: the same parity as kx, so (SPLIT) it is in this wavefront's half, at k' = NF - k - h (mod NF)
          const float2 cm = SPLIT ? fbuf[(NF - (tf + m * TF) - h) & (NF - 1)] : buf[(N - kx) & (N - 1)];
          const float2 f0 = make_float2(0.5f * (ck.x + cm.x), 0.5f * (ck.y - cm.y));
          const float2 fn = make_float2(0.5f * (ck.y + cm.y), -0.5f * (ck.x - cm.x));
          const float q0 = amp_to_q<LOG>(f0), qn = amp_to_q<LOG>(fn);
          a1 += w[m].x * q0;
          a2 += w[m].x * q0 * q0;
          n1 += wnm.x * qn;
          n2 += wnm.x * qn * qn;
          if constexpr (EPI == EPI_QSTORE) {
            if constexpr (T <= 64) {
              if (compact) {   // row N/2's weights arrive per candidate: its positions with them (one wavefront in 256)
                const unsigned long long bal = __ballot(wnm.x > 0.f);
                float* const qb = a.q_out + b * a.q_stride;
                const unsigned long long bal0 = __ballot(w[m].x > 0.f);
                if (w[m].x > 0.f) qb[zbase + __popcll(bal0 & ((1ull << t) - 1ull))] = q0;
                zbase += __popcll(bal0);
                if (wnm.x > 0.f) qb[nbase + __popcll(bal & ((1ull << t) - 1ull))] = qn;
                nbase += __popcll(bal);
              }
            }
            if (!compact) {
              float* const q0row = a.q_out + b * (size_t)(N / 2 + 1) * N;   // (lane-major rows: kx = wl + m T at [wl][m])
              q0row[wl * 8 + m] = w[m].x > 0.f ? q0 : 0.f;
              q0row[(size_t)(N / 2) * N + wl * 8 + m] = wnm.x > 0.f ? qn : 0.f;
            }
          } else {
            a3 += w[m].y * q0;
            n3 += wnm.y * qn;
          }
        }
        group_sum3<TL>(a1, a2, a3);
        group_sum3<TL>(n1, n2, n3);
        if (writer) {
          double* const o = a.partials + (b * KP::NPART + wave_in_row) * 3;
          o[0] = a1;
          o[1] = a2;
          o[2] = a3;
          double* const on = a.partials + (b * KP::NPART + (size_t)KP::ROWS * KP::WPR + wave_in_row) * 3;
          on[0] = n1;
          on[1] = n2;
          on[2] = n3;
        }
        scored = true;
      }
    }
    if (!scored) {
      float s1 = 0.f, s2 = 0.f, s3 = 0.f;
      float* const qrow = (EPI == EPI_QSTORE) ? a.q_out + (b * (size_t)(N / 2 + 1) + row) * N : nullptr;
      [[maybe_unused]] float qkeep[8];
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const float q = (HH_ABLATE & 32) ? v[m].x + v[m].y : amp_to_q<LOG>(v[m]);
        s1 += w[m].x * q;
        s2 += w[m].x * q * q;
        if constexpr (EPI == EPI_QSTORE) {
          if (defer_q || compact) {
            qkeep[m] = q;   // compact: stored below / deferred: at the top of the next round (flush_q)
          } else {
            qrow[wl * 8 + m] = w[m].x > 0.f ? q : 0.f;   // lane-major: two 16-byte stores per lane and row
          }
        } else {
          s3 += w[m].y * q;
        }
      }
      if constexpr (EPI == EPI_QSTORE) {
        if (defer_q) {   // the row's q into the group's own exchange buffer (free until the next candidate's panel)
          float4* const st = reinterpret_cast<float4*>(fbuf) + 2 * tf;
          st[0] = make_float4(qkeep[0], qkeep[1], qkeep[2], qkeep[3]);
          st[1] = make_float4(qkeep[4], qkeep[5], qkeep[6], qkeep[7]);
        } else if (compact) {
          store_q_compact(b, qkeep);   // only the bins with weight: a fifth fewer bytes under the radial band
        }
      }
      group_sum3<TL>(s1, s2, s3);
      if (writer) {
        double* const o = a.partials + (b * KP::NPART + (size_t)row * KP::WPR + wave_in_row) * 3;
        o[0] = s1;
        o[1] = s2;
        o[2] = s3;
      }
    }
  };
  // Several segments: the q stores of a candidate must not sit right in front of the round's s_waitcnt vmcnt(0) (it retires
  // the LDS-DMA copies, and on this chip stores count on vmcnt too): the wavefronts that run A then B issued them last and then
  // waited out their whole latency, every round — the fused pass with q stores took 7.0 ms per 20,000 candidates against 3.9
  // without.  Those wavefronts park the row's q in their exchange buffer and store it at the top of the NEXT round, so the
  // stores have a round to land; the late wavefronts (B of the previous candidate first) had that order already.
  auto flush_q = [&](int cc) {
    if constexpr (EPI == EPI_QSTORE) {
      const size_t b = (size_t)(cfirst + cc);
      const float4* const st = reinterpret_cast<const float4*>(fbuf) + 2 * tf;
      const float4 q0 = st[0], q1 = st[1];
      const float qv[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
      if (compact) {
        store_q_compact(b, qv);
      } else {
        float* const qrow = a.q_out + (b * (size_t)(N / 2 + 1) + row) * N;
#pragma unroll
        for (int m = 0; m < 8; ++m) qrow[wl * 8 + m] = w[m].x > 0.f ? qv[m] : 0.f;
      }
    }
  };

#pragma unroll 1
  for (int it = 0; it < nc; ++it) {
    if (defer_q && it > 0) flush_q(it - 1);
    // next candidate's column factors: copied global -> LDS by the load unit itself (no registers, no ds_write
    // pass) into the other factor buffer, which nobody reads before the barrier at the end of this round
    // (an explicit s_waitcnt vmcnt(0) ahead of that barrier retires the copies).  One wave-instruction moves
    // 64 x 16 B to a wave-uniform LDS base + lane x 16.
    const bool more = it + 1 < nc && !(HH_ABLATE & 4096) && !EGLOBAL;
    if (more) {
      const size_t bn = (size_t)(cfirst + it + 1);
      const char* const gsrc = reinterpret_cast<const char*>(a.eg + bn * a.kg * N);
      char* const ldst = reinterpret_cast<char*>(eg + (size_t)((it & 1) ^ 1) * a.kg * N);
      const int lane = tid & 63, wave = tid >> 6;
      for (int p0 = wave * 64; p0 < n_e4; p0 += K::THREADS)  // (a workgroup narrower than a wavefront: wave = 0)
        if (p0 + lane < n_e4) lds_dma16(gsrc + (size_t)(p0 + lane) * 16, lds_offset_of(ldst + (size_t)p0 * 16));
      // the groups' first table rows and the row count: N/4 + 4 ints = CGS / 4 pieces (65 at N = 1024: two wavefronts)
      for (int p0 = wave * 64; p0 < CGS / 4; p0 += K::THREADS)
        if (p0 + lane < CGS / 4)
          lds_dma16(reinterpret_cast<const char*>(a.cgs + bn * CGS) + (size_t)(p0 + lane) * 16,
                    lds_offset_of(reinterpret_cast<const char*>(cgs + ((it & 1) ^ 1) * CGS) + (size_t)p0 * 16));
    }
    if (late && it > 0) part_b(it - 1);
    part_a(it);
    if (!late) part_b(it);
    // The LDS-DMA copies of the next candidate's factors count on vmcnt only; neither the workgroup-scope fence nor
    // s_barrier waits for them, so every wavefront retires its own copies before it arrives at the barrier.
    if (more && !(HH_ABLATE & 32768)) lds_dma_wait();
    if (!(HH_ABLATE & 16384) && !EGLOBAL) __syncthreads();  // the next candidate's factors are complete; every group is done reading this one's
  }
  if (late) part_b(nc - 1);
  if (defer_q) flush_q(nc - 1);
}

// ------------------------------------------------------------------------------------------
// Several experimental segments against one candidate grid (BASELINE config 5): the covariance
// numerators S3[s][c] = sum_k WEC[s][k] * Q[c][k] are a dense contraction over the K = (N/2+1) N
// half-plane bins, so they run on the matrix cores with the exact-f32 MFMA (32x32x2, f32 in /
// f32 accumulate): one wavefront owns a 64-candidate x 64-segment tile of ONE spectrum row (N bins)
// and streams its operands straight from L2 / Infinity Cache into registers, 16 B per lane.
// ------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

// A workgroup of four wavefronts owns up to 256 candidates x 64 segments of ONE spectrum row (n_k bins):
// the operands are staged through LDS in slices of 32 bins with whole-line global loads (eight lanes per
// 128-byte line; reading them straight from global put every lane of a load on a different row and
// thrashed the L1), the next slice's loads fly under the current slice's MFMAs.
constexpr int SC_KC = 32;            // bins per staged slice
// LDS row stride in floats.  The MFMA operand of lane (i, h) is element [i][kk + h]: 32 rows read at once, so the stride
// must spread 32 rows over 32 different pairs of banks — 34 does (34 i mod 64 = 2 (17 i mod 32)), 36 = the first
// 16-byte-aligned choice does not (36 i mod 64 takes 16 values: two lanes per bank, SQ_LDS_BANK_CONFLICT 64 % of the LDS
// cycles and the matrix pipe 58 % busy).  Rows are 8-byte aligned, so a staged float4 goes in as two float2 (a write
// group of 16 lanes = two rows, whose bank offsets 0 and 34 mod 32 = 2 interleave).
constexpr int SC_LD = SC_KC + 2;

__global__ __launch_bounds__(256) void k_segment_corr(const float* __restrict__ q /*[Bp][K]*/,
                                                      const float* __restrict__ wec /*[Sp][K]*/, int n_k /*N*/,
                                                      size_t K, int Bp, int Sp, float* __restrict__ part /*[rows][Bp][Sp]*/,
                                                      const int* __restrict__ slices /*[gridDim.x] or NULL*/) {
  __shared__ __attribute__((aligned(16))) float sA[256 * SC_LD];
  __shared__ __attribute__((aligned(16))) float sB[64 * SC_LD];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 31, h = lane >> 5;
  // spectrum row = K-slice of n_k bins; with `slices` only the listed ones (those the mask gives any weight) are
  // contracted, and part is indexed by the position in the list
  const int row = slices ? slices[blockIdx.x] : (int)blockIdx.x;
  const int c0 = blockIdx.y * 256, s0 = blockIdx.z * 64;
  const int lr = tid >> 3, lc = (tid & 7) * 4;  // staging: thread covers 16 bytes of rows lr + 32 j
  const int na = min(256, Bp - c0);             // candidate rows that exist (Bp is a multiple of 64)
  const float* const qbase = q + (size_t)c0 * K + (size_t)row * n_k + lc;
  const float* const wbase = wec + (size_t)s0 * K + (size_t)row * n_k + lc;
  float4 ra[8], rb[2];
  auto fetch = [&](int k) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = lr + 32 * j;
      ra[j] = r < na ? *reinterpret_cast<const float4*>(qbase + (size_t)r * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) rb[j] = *reinterpret_cast<const float4*>(wbase + (size_t)(lr + 32 * j) * K + k);
  };
  f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
  fetch(0);
  for (int k = 0; k < n_k; k += SC_KC) {
    __syncthreads();  // the previous slice has been consumed
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float2* const d = reinterpret_cast<float2*>(sA + (lr + 32 * j) * SC_LD + lc);
      d[0] = make_float2(ra[j].x, ra[j].y);
      d[1] = make_float2(ra[j].z, ra[j].w);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float2* const d = reinterpret_cast<float2*>(sB + (lr + 32 * j) * SC_LD + lc);
      d[0] = make_float2(rb[j].x, rb[j].y);
      d[1] = make_float2(rb[j].z, rb[j].w);
    }
    __syncthreads();
    if (k + SC_KC < n_k) fetch(k + SC_KC);
    const float* const a0p = sA + (wave * 64 + i) * SC_LD + 2 * h;
    const float* const a1p = a0p + 32 * SC_LD;
    const float* const b0p = sB + i * SC_LD + 2 * h;
    const float* const b1p = b0p + 32 * SC_LD;
    // One 8-byte read feeds two MFMA steps: lane (i, h) supplies bins kk + 2 h (first step) and kk + 2 h + 1 (second) of
    // row i for A and for B alike — both operands use the same k numbering, which is all a dot product needs.  With the
    // row stride 34 the 32 lanes of a read group land on 32 different bank pairs (ds_read_b64 banks are mod 64; the
    // 4-byte reads this replaces bank mod 32 and were two-way conflicts at any even stride).
#pragma unroll
    for (int kk = 0; kk < SC_KC; kk += 4) {
      const float2 a0 = *reinterpret_cast<const float2*>(a0p + kk), a1 = *reinterpret_cast<const float2*>(a1p + kk);
      const float2 b0 = *reinterpret_cast<const float2*>(b0p + kk), b1 = *reinterpret_cast<const float2*>(b1p + kk);
      acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc00, 0, 0, 0);
      acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b1.x, acc01, 0, 0, 0);
      acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b0.x, acc10, 0, 0, 0);
      acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b1.x, acc11, 0, 0, 0);
      acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc00, 0, 0, 0);
      acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b1.y, acc01, 0, 0, 0);
      acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b0.y, acc10, 0, 0, 0);
      acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1.y, acc11, 0, 0, 0);
    }
  }
  // D layout of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  const int cw = c0 + wave * 64;
  if (cw >= Bp) return;
  float* const out = part + (size_t)blockIdx.x * Bp * Sp;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int ci = (r & 3) + 8 * (r >> 2) + 4 * h;
    out[(size_t)(cw + ci) * Sp + s0 + i] = acc00[r];
    out[(size_t)(cw + ci) * Sp + s0 + 32 + i] = acc01[r];
    out[(size_t)(cw + 32 + ci) * Sp + s0 + i] = acc10[r];
    out[(size_t)(cw + 32 + ci) * Sp + s0 + 32 + i] = acc11[r];
  }
}

// scores[s][g0 + c] for one batch: moments s1, s2 from K_B's partials, s3 summed over spectrum rows
__global__ void k_finalize_segments(const double* __restrict__ partials, int nblk, const float* __restrict__ part,
                                    int rows, int Bp, int Sp, int nb, int n_seg, const RefConsts* __restrict__ rc,
                                    float* __restrict__ scores, int64_t g, int64_t g0) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nb * n_seg) return;
  const int cand = idx / n_seg, s = idx % n_seg;
  double s1 = 0, s2 = 0, s3 = 0;
  for (int k = 0; k < nblk; ++k) {
    s1 += partials[((size_t)cand * nblk + k) * 3];
    s2 += partials[((size_t)cand * nblk + k) * 3 + 1];
  }
  {
    // rows independent loads in flight (a serial loop of dependent waits cost 60 us per batch)
    const float* const p0 = part + (size_t)cand * Sp + s;
    const size_t stride = (size_t)Bp * Sp;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    int r = 0;
    for (; r + 8 <= rows; r += 8) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = p0[(size_t)(r + j) * stride];
      a0 += (double)v[0] + (double)v[4];
      a1 += (double)v[1] + (double)v[5];
      a2 += (double)v[2] + (double)v[6];
      a3 += (double)v[3] + (double)v[7];
    }
    for (; r < rows; ++r) a0 += (double)p0[(size_t)r * stride];
    s3 = (a0 + a1) + (a2 + a3);
  }
  const RefConsts c = rc[s];
  double score = 0.0;
  if (c.sw > 0) {
    const double var_q = s2 - s1 * s1 / c.sw;
    const double cov = s3 - (s1 / c.sw) * c.swec;
    const double den = var_q * c.var_e;
    if (den > 0 && var_q > 1e-9 * s2) score = cov / sqrt(den);
  }
  scores[(size_t)s * g + g0 + cand] = (float)score;
}

// ------------------------------------------------------------------------------------------
// spectrum expansion for hh_power_spectrum, and the vector reductions
// ------------------------------------------------------------------------------------------
__global__ void k_expand_spectrum(const float2* __restrict__ spec, int n, int log_flag, float* __restrict__ pwr,
                                  float* __restrict__ phase, unsigned* __restrict__ minmax) {
  const int iy = blockIdx.x;
  const int uy = (iy + n / 2) & (n - 1);  // unshifted frequency index
  float lmin = INFINITY, lmax = 0.f;
  for (int ix = threadIdx.x; ix < n; ix += blockDim.x) {
    const int ux = (ix + n / 2) & (n - 1);
    float2 f;
    if (uy <= n / 2) {
      f = spec[(size_t)uy * n + ux];
    } else {
      f = spec[(size_t)(n - uy) * n + ((n - ux) & (n - 1))];
      f.y = -f.y;
    }
    const float a = sqrtf(f.x * f.x + f.y * f.y);
    const float q = log_flag ? log1pf(a) : a;
    pwr[(size_t)iy * n + ix] = q;
    if (phase) phase[(size_t)iy * n + ix] = atan2f(f.y, f.x);
    lmin = fminf(lmin, q);
    lmax = fmaxf(lmax, q);
  }
  // q >= 0, so the IEEE bit patterns order like unsigned integers
  atomicMin(&minmax[0], __float_as_uint(lmin));
  atomicMax(&minmax[1], __float_as_uint(lmax));
}

__global__ void k_normalise(float* __restrict__ pwr, size_t n, const unsigned* __restrict__ minmax) {
  const float vmin = __uint_as_float(minmax[0]), vmax = __uint_as_float(minmax[1]);
  if (vmax == vmin) return;  // filters.py:278-279
  const float inv = 1.0f / (vmax - vmin);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    pwr[i] = (pwr[i] - vmin) * inv;
}

// arg-max of every row of scores[rows][n] (np.argmax: lowest index on ties; NaN never wins; all-NaN -> 0)
__global__ __launch_bounds__(1024) void k_argmax_rows(const float* __restrict__ scores, int64_t n, int64_t ld,
                                                      int64_t* __restrict__ out) {
  const float* const row = scores + (size_t)blockIdx.x * ld;
  float bv = -INFINITY;
  int64_t bi = -1;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
    const float v = row[i];
    if (v != v) continue;
    if (bi < 0 || v > bv) { bv = v; bi = i; }  // ascending i per thread: the first maximum is kept
  }
  auto better = [](float v, int64_t i, float bv, int64_t bi) { return i >= 0 && (bi < 0 || v > bv || (v == bv && i < bi)); };
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_down(bv, off, 64);
    const int64_t oi = __shfl_down(bi, off, 64);
    if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
  }
  __shared__ float sv[16];
  __shared__ int64_t si[16];
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sv[w] = bv; si[w] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < (int)(blockDim.x >> 6); ++k)
      if (better(sv[k], si[k], bv, bi)) { bv = sv[k]; bi = si[k]; }
    out[blockIdx.x] = bi < 0 ? 0 : bi;
  }
}

template <typename T>
__global__ void k_pair_sums(const T* __restrict__ x, const T* __restrict__ y, int64_t n, double mx, double my,
                            double* __restrict__ out /*[grid][5]*/) {
  double sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double p = (double)x[i] - mx, q = (double)y[i] - my;
    sx += p;
    sy += q;
    sxx += p * p;
    syy += q * q;
    sxy += p * q;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    sx += __shfl_down(sx, off, 64);
    sy += __shfl_down(sy, off, 64);
    sxx += __shfl_down(sxx, off, 64);
    syy += __shfl_down(syy, off, 64);
    sxy += __shfl_down(sxy, off, 64);
  }
  __shared__ double red[4][5];
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[w][0] = sx; red[w][1] = sy; red[w][2] = sxx; red[w][3] = syy; red[w][4] = sxy;
  }
  __syncthreads();
  if (threadIdx.x < 5) {
    double r = 0;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) r += red[k][threadIdx.x];
    out[blockIdx.x * 5 + threadIdx.x] = r;
  }
}

// ------------------------------------------------------------------------------------------
// Pre-sweep image preparation: Gaussian low / high pass in Fourier space
// [filters.py:314-372] and threshold_data [filters.py:283-311].  The forward transform is the sweep's
// own (k_first_pass<MODE_IMAGE> + k_second_pass<EPI_STORE>: half-plane spectrum [N/2+1][N]); the two
// kernels below multiply by the filter and transform back: rows along x (complex), then columns along y
// with two real output columns packed into one complex transform (the inverse of the first pass's trick).
// An inverse transform is the forward one on conjugated data, conjugated again.
// ------------------------------------------------------------------------------------------
// filter value at unshifted frequency indices (ky, kx) [filters.py:343-369: the fftshifted meshgrid in
// float32, then exp(-f2 R2) and 1 - exp(-f2 R2)]
__device__ __forceinline__ float pass_filter(int ky, int kx, int n, float f2_lp, float f2_hp) {
  const int fy = ky < n / 2 ? ky : ky - n, fx = kx < n / 2 ? kx : kx - n;
  const float y = (float)fy / (float)(n / 2), x = (float)fx / (float)(n / 2);
  const float r2 = x * x + y * y;
  float f = 1.f;
  if (f2_lp > 0.f) f *= expf(-f2_lp * r2);
  if (f2_hp > 0.f) f *= 1.0f - expf(-f2_hp * r2);
  return f;
}

// one workgroup = one transform: row ky of the half-plane spectrum, filtered, inverse-transformed along x
template <int N>
__global__ __launch_bounds__(N / 8) void k_filter_rows(const float2* __restrict__ spec, const float2* __restrict__ twtab,
                                                        float f2_lp, float f2_hp, float2* __restrict__ out) {
  constexpr int T = N / 8;
  __shared__ __attribute__((aligned(16))) float2 buf[N];
  const int t = threadIdx.x, ky = blockIdx.x;
  float2 tw[TwN<N>::total];
  load_twiddles<N>(tw, t, twtab);
  const TwRegs twsrc{tw};
  float2 v[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const int kx = t + m * T;
    const float2 f = spec[(size_t)ky * N + kx];
    const float w = pass_filter(ky, kx, N, f2_lp, f2_hp);
    v[m] = make_float2(w * f.x, -w * f.y);  // conj
  }
  fft_lanes<N>(v, twsrc, t, buf);
#pragma unroll
  for (int m = 0; m < 8; ++m) out[(size_t)ky * N + t + m * T] = make_float2(v[m].x, -v[m].y);  // conj back (unscaled)
}

// one workgroup = one pair of image columns (xa, xa + 1): Z[ky] = Ga[ky] + i Gb[ky] over all N ky
// (Hermitian extension of the half plane), inverse transform along y, a = Re z, b = Im z, scaled by 1/N^2
template <int N>
__global__ __launch_bounds__(N / 8) void k_inverse_cols(const float2* __restrict__ g, const float2* __restrict__ twtab,
                                                         float* __restrict__ image) {
  constexpr int T = N / 8;
  __shared__ __attribute__((aligned(16))) float2 buf[N];
  const int t = threadIdx.x, xa = 2 * blockIdx.x;
  float2 tw[TwN<N>::total];
  load_twiddles<N>(tw, t, twtab);
  const TwRegs twsrc{tw};
  float2 v[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const int ky = t + m * T;
    const bool mirror = ky > N / 2;
    const int src = mirror ? N - ky : ky;
    const float4 ab = *reinterpret_cast<const float4*>(g + (size_t)src * N + xa);  // Ga, Gb of row src
    // mirrored rows: G[ky][x] = conj(G[N - ky][x]) (the filtered image is real)
    const float2 ga = make_float2(ab.x, mirror ? -ab.y : ab.y), gb = make_float2(ab.z, mirror ? -ab.w : ab.w);
    const float2 z = make_float2(ga.x - gb.y, ga.y + gb.x);  // Ga + i Gb
    v[m] = make_float2(z.x, -z.y);                            // conj
  }
  fft_lanes<N>(v, twsrc, t, buf);
  const float scale = 1.0f / ((float)N * (float)N);
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const int y = t + m * T;
    *reinterpret_cast<float2*>(image + (size_t)y * N + xa) = make_float2(v[m].x * scale, -v[m].y * scale);
  }
}

// threshold_data: max over the array (non-negative-safe float ordering via two atomics), then clip - thresh
__global__ void k_array_max(const float* __restrict__ x, size_t n, int* __restrict__ out /*[2]: max of >=0 as int, min of <0 as uint*/) {
  float m = -INFINITY;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) m = fmaxf(m, x[i]);
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off, 64));
  if ((threadIdx.x & 63) == 0) {
    // float bits order like signed ints for non-negative values and reversed for negative ones
    if (m >= 0.f) atomicMax(&out[0], __float_as_int(m));
    else atomicMin(reinterpret_cast<unsigned*>(&out[1]), __float_as_uint(m));
  }
}

__global__ void k_threshold(const float* __restrict__ x, size_t n, const int* __restrict__ mx, float fraction, float value,
                            int use_fraction, float* __restrict__ out) {
  float thresh = value;
  if (use_fraction) {
    const float vmax = mx[0] >= 0 ? __int_as_float(mx[0]) : __uint_as_float((unsigned)mx[1]);
    thresh = vmax * fraction;
  }
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = fmaxf(x[i], thresh) - thresh;  // np.clip(data, thresh, None) - thresh
}

// ------------------------------------------------------------------------------------------
// apply_helical_symmetry (reference: lib/transforms.py:58-165): every output voxel gathers, for
// each helical repeat hi in [-hmax, hmax] and cyclic copy ci, the trilinearly interpolated input
// at the symmetry-related position and averages.  One thread per output voxel; coordinates and
// weights in float64 and the running sum rounded to float32 after every term, in the reference's
// (hi, ci) order, with FP contraction off, so the result matches NumPy bit for bit up to the
// last-ulp difference of the host's cos/sin.
// ------------------------------------------------------------------------------------------
struct SymArgs {
  const float* data;   // [nz0][ny0][nx0]
  float* out;          // [oz][oy][ox]
  const double* rot;   // [(2 hmax + 1) * csym][2] = cos, sin of twist * hi + 360 ci / csym
  int nz0, ny0, nx0;   // input volume
  int nz, ny, nx;      // work volume = element-wise max of input and requested size
  int cz, cy, cx;      // crop origin of the output inside the work volume
  int oz, oy, ox;      // output size
  int hmax, csym, z0, z1;
  double apix, new_apix, rise;
};

__global__ __launch_bounds__(256) void k_apply_helical_symmetry(SymArgs a) {
#pragma clang fp contract(off)
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)a.oz * a.oy * a.ox;
  if (idx >= total) return;
  const int io = (int)(idx % a.ox), jo = (int)((idx / a.ox) % a.oy), ko = (int)(idx / ((size_t)a.ox * a.oy));
  const int k = ko + a.cz, j = jo + a.cy, i = io + a.cx;
  const double jj = (double)(j - a.ny / 2);   // integer centre (transforms.py:117)
  const double ii = (double)i - (double)a.nx / 2.0;  // float centre: the reference's nx / 2
  float acc = 0.f, cnt = 0.f;
  for (int hi = -a.hmax; hi <= a.hmax; ++hi) {
    const double k2 = ((double)(k - a.nz / 2) * a.new_apix + (double)hi * a.rise) / a.apix + (double)(a.nz0 / 2);
    if (k2 < (double)a.z0 || k2 >= (double)a.z1) continue;
    const int kf = (int)floor(k2), kc = (int)ceil(k2);
    const double wk = k2 - (double)kf;
    const float* const d0 = a.data + (size_t)kf * a.ny0 * a.nx0;
    const float* const d1 = a.data + (size_t)kc * a.ny0 * a.nx0;
    for (int ci = 0; ci < a.csym; ++ci) {
      const double c = a.rot[2 * ((size_t)(hi + a.hmax) * a.csym + ci)];
      const double s = a.rot[2 * ((size_t)(hi + a.hmax) * a.csym + ci) + 1];
      const double j2 = (c * jj + s * ii) * a.new_apix / a.apix + (double)(a.ny0 / 2);
      const double i2 = ((-s) * jj + c * ii) * a.new_apix / a.apix + (double)(a.nx0 / 2);
      const double j2f = floor(j2), i2f = floor(i2);
      if (j2f < 0.0 || j2f >= (double)(a.ny0 - 1) || i2f < 0.0 || i2f >= (double)(a.nx0 - 1)) continue;
      const int jf = (int)j2f, jc = (int)ceil(j2), i_f = (int)i2f, ic = (int)ceil(i2);
      const double wj = j2 - j2f, wi = i2 - i2f;
      double v = (1 - wk) * (1 - wj) * (1 - wi) * (double)d0[(size_t)jf * a.nx0 + i_f];
      v = v + (1 - wk) * (1 - wj) * wi * (double)d0[(size_t)jf * a.nx0 + ic];
      v = v + (1 - wk) * wj * (1 - wi) * (double)d0[(size_t)jc * a.nx0 + i_f];
      v = v + (1 - wk) * wj * wi * (double)d0[(size_t)jc * a.nx0 + ic];
      v = v + wk * (1 - wj) * (1 - wi) * (double)d1[(size_t)jf * a.nx0 + i_f];
      v = v + wk * (1 - wj) * wi * (double)d1[(size_t)jf * a.nx0 + ic];
      v = v + wk * wj * (1 - wi) * (double)d1[(size_t)jc * a.nx0 + i_f];
      v = v + wk * wj * wi * (double)d1[(size_t)jc * a.nx0 + ic];
      acc = (float)((double)acc + v);  // data_work is float32 in the reference (transforms.py:81-83, 133)
      cnt += 1.0f;
    }
  }
  a.out[idx] = cnt > 0.f ? acc / cnt : acc;
}

// ------------------------------------------------------------------------------------------
// traffic-counter calibration (profiling aid): the sweep's two global access shapes on a known
// byte count, so rocprofv3's FETCH_SIZE / WRITE_SIZE can be turned into bytes for THESE shapes
// (MI355X_MICROARCH.md, HBM: FETCH_SIZE is only calibrated for 16-B-per-lane streams)
// ------------------------------------------------------------------------------------------
// mode 0: K_B's read shape — a workgroup streams 32 KB ky blocks, 16 B per lane, contiguous
__global__ __launch_bounds__(512) void k_calib_read(const float4* __restrict__ src, size_t blocks, float* __restrict__ sink) {
  float acc = 0.f;
  for (size_t kb = blockIdx.x; kb < blocks; kb += gridDim.x) {
    const float4* p = src + kb * 2048 + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 x = p[i * 512];
      acc += x.x + x.y + x.z + x.w;
    }
  }
  if (acc == 123.456f) sink[0] = acc;  // keeps the loads alive, never true for the zero-filled buffer
}

// mode 1: K_A's write shape — each wavefront writes whole 128-byte lines (8 lanes x 16 B), the
// lines of one store instruction 32 KB apart (one per ky block), as k_first_pass<512> does
__global__ __launch_bounds__(512) void k_calib_write(float2* __restrict__ dst, size_t cands) {
  const int t = threadIdx.x & 63, f = threadIdx.x >> 6;
  for (size_t job = blockIdx.x; job < cands * 32; job += gridDim.x) {  // 32 tiles of 16 columns per candidate
    float2* const out = dst + (job / 32) * (size_t)256 * 512;
    const int pair = (int)(job % 32) * 8 + f;
#pragma unroll
    for (int m = 0; m < 4; ++m)
      *reinterpret_cast<float4*>(out + inter_index<512>(t + 64 * m, pair)) = make_float4(1.f, 2.f, 3.f, 4.f);
  }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
thread_local std::string g_create_error;

struct EventPair {
  hipEvent_t a, b;
  int kind;
};

}  // namespace

#include "general_sizes.inc"  // kernels of the general-size path (any ny x nx) and struct hh_gen
#include "path_a.inc"         // Path A (sparse least-squares scorer): projector kernels and struct hh_pa

struct hh_ctx {
  int device = 0;
  int n = 0;                     // side of a square power-of-two context; 0 for a general-size one
  int ny = 0, nx = 0;            // image rows / columns (the helical axis runs along the columns)
  bool general = false;          // true: every entry point goes through general_sizes.inc / general_host.inc
  void* comm = nullptr;          // ncclComm_t of hh_comm_init (RCCL, loaded on first use)
  hh_gen* gen = nullptr;
  int max_batch = 0;
  int n_cu = 0;                  // compute units of the device
  int slots = 0;                 // resident k_fused_pass workgroups for the launch shape slots_key (fused_slots)
  size_t slots_key = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::string err;

  float2* d_tw = nullptr;
  float2* d_inter = nullptr;     // [max_batch][N/2][N]
  double* d_partials = nullptr;  // [2][max_batch][NPART][3], zero where the mask skips a ky block; the halves alternate
                                 // between batches (the fused pass scores batch i-1 while it writes batch i)
  double* d_psum = nullptr;      // S > 1: [b_pad][3] summed partials
  double* d_params = nullptr;    // staging for hh_sweep
  float* d_scores = nullptr;
  int64_t cap_params = 0, cap_scores = 0;
  double* d_units = nullptr;
  float2* d_w2 = nullptr;        // [N/2+1][N/8][8] (lane-major within a row); with S > 1 only w is used
  float* d_wec = nullptr;        // S > 1: [Sp][K] w (E_s - Ebar_s), rows in q's lane-major bin order, Sp = S rounded up to 64
  float* d_q = nullptr;          // S > 1: [Bp][K] masked q of one batch, Bp = max_batch rounded up to 64
  float* d_cpart = nullptr;      // S > 1: [N/2+1][Bp][Sp] per-row covariance numerators
  RefConsts* d_ref = nullptr;    // S > 1: [S]
  int64_t* d_argmax = nullptr;   // hh_argmax_device: one index per row
  size_t cap_argmax = 0;
  size_t cap_w2 = 0, cap_wec = 0, cap_q = 0, cap_cpart = 0, cap_ref = 0, cap_psum = 0, cap_kb = 0;  // bytes
  int* d_kb_list = nullptr;      // ky blocks (8 rows) the mask touches, ascending; block 0 always
  int* d_q_roff = nullptr;       // S > 1, N <= 512: compact q — first position of every spectrum row's weighted bins
  size_t cap_q_roff = 0;
  size_t q_stride = 0;           // S > 1: floats per candidate in d_q (compact: weighted bins, rounded up to 512; else (N/2+1) N)
  int q_slices = 0;              // compact: 512-bin slices of the flat axis the contraction walks
  int* d_seg_rows = nullptr;     // S > 1: spectrum rows with any weight, ascending (the contraction skips the others)
  size_t cap_seg_rows = 0;
  int n_seg_rows = 0;
  int n_kb = 0;
  float2* d_table = nullptr;     // shared-twist first pass: [runs per batch][rows][N/2] column-transform table
  size_t cap_table = 0;          // bytes
  int* d_run_imax = nullptr;     // [runs of the sweep]
  int64_t cap_runs = 0;
  std::vector<int> h_run_imax;
  int table_path = 1;            // 0: never take the shared-twist first pass
  int fused_path = 1;            // 0: shared-twist runs go through the two-pass pipeline (k_first_pass_table + k_second_pass)
  float* d_eg = nullptr;         // fused pass: [max_batch][kg][N] column factors
  int* d_cgs = nullptr;          // fused pass: [max_batch][N/4]
  size_t cap_eg = 0, cap_cgs = 0;
  int cap_partials = 0;          // candidates per half of d_partials
  int last_first_pass = 0;       // what the last sweep ran: 0 per-candidate transform, 1 run tables
  unsigned long long kb_mask = ~0ull;
  int s_pad = 0, b_pad = 0;
  float2* d_spec = nullptr;      // [N/2+1][N] scratch (grown for S segments)
  int64_t cap_spec = 0;
  float* d_img = nullptr;        // scratch images
  int64_t cap_img = 0;
  std::vector<RefConsts> ref;
  int n_segments = 0;
  int log_flag = 1;
  bool have_geom = false;
  DevGeom geom{};

  // kernels whose dynamic-LDS limit has been raised on THIS context's device (function -> bytes granted); the
  // attribute is per device, so it is tracked per context, never per process
  std::vector<std::pair<const void*, int>> lds_attr;

  int profiling = 0;             // 0 off, k > 0: time every k-th batch of a sweep
  bool prof_now = false;         // the batch being enqueued is a sampled one
  std::vector<EventPair> events;
  size_t events_used = 0;
  int64_t prof_candidates = 0;
};

namespace {

// general-size path (general_host.inc, included at the end of this namespace)
int gen_sweep(hh_ctx* c, const double* d_params, const double* h_params, int64_t n_cand, float* d_scores, int64_t ld);

int fail(hh_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg; else g_create_error = msg;
  return code;
}

// Exception barrier of the C ABI ("never exit()").  The host code under the entry points uses
// std::vector / std::string / std::thread; an exception that left an extern "C" function would end the caller's
// process in std::terminate (a Python process, through ctypes).  Every entry point with a body of its own is a
// function-try-block whose handler comes here: the exception in flight is rethrown and sorted into a status code and a
// message for hh_*_last_error.  `set` is fail / pab_fail / pa_fail bound to the object of the call.
template <class Set>
int hh_caught(Set&& set, const char* fn) noexcept {
  try {
    try {
      throw;
    } catch (const std::bad_alloc&) {
      return set(HH_ERR_NOMEM, std::string(fn) + ": out of host memory (std::bad_alloc)");
    } catch (const std::length_error& e) {   // a container asked for more elements than it can address: an absurd count
      return set(HH_ERR_NOMEM, std::string(fn) + ": " + e.what() + " (std::length_error)");
    } catch (const std::exception& e) {
      return set(HH_ERR_INTERNAL, std::string(fn) + ": " + e.what());
    } catch (...) {
      return set(HH_ERR_INTERNAL, std::string(fn) + ": unknown C++ exception");
    }
  } catch (...) {   // even the message could not be built
    return HH_ERR_NOMEM;
  }
}
#define HH_CATCH_CTX(c, fn) catch (...) { return hh_caught([&](int code, const std::string& m) { return fail(const_cast<hh_ctx*>(static_cast<const hh_ctx*>(c)), code, m); }, fn); }
#define HH_CATCH_PAB(p, fn) catch (...) { return hh_caught([&](int code, const std::string& m) { return pab_fail(const_cast<hh_pab*>(static_cast<const hh_pab*>(p)), code, m); }, fn); }
#define HH_CATCH_PA(p, fn) catch (...) { return hh_caught([&](int code, const std::string& m) { return pa_fail(const_cast<hh_pa*>(static_cast<const hh_pa*>(p)), code, m); }, fn); }

#define HH_HIP(ctx, call)                                                                          \
  do {                                                                                             \
    hipError_t e__ = (call);                                                                       \
    if (e__ != hipSuccess)                                                                         \
      return fail(ctx, HH_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__));            \
  } while (0)

bool supported_n(int n) { return n == 32 || n == 64 || n == 128 || n == 256 || n == 512 || n == 1024; }

int npart_for(int n) {  // partial-moment slots per candidate (KB<N>::NPART)
  switch (n) {
    case 32: return KB<32>::NPART;
    case 64: return KB<64>::NPART;
    case 128: return KB<128>::NPART;
    case 256: return KB<256>::NPART;
    case 512: return KB<512>::NPART;
    default: return KB<1024>::NPART;
  }
}

#define HH_SWITCH_N(c, CALL)                                        \
  switch ((c)->n) {                                                 \
    case 32: return CALL(32);                                       \
    case 64: return CALL(64);                                       \
    case 128: return CALL(128);                                     \
    case 256: return CALL(256);                                     \
    case 512: return CALL(512);                                     \
    case 1024: return CALL(1024);                                   \
  }                                                                 \
  return fail(c, HH_ERR_ARG, "unsupported image size")

// hipFuncAttributeMaxDynamicSharedMemorySize applies to the device that is current when it is set; a context is
// bound to one device, so it remembers what it has set (the caller has made c->device current).
int ensure_lds_attr(hh_ctx* c, const void* fn, int bytes) {
  for (auto& e : c->lds_attr)
    if (e.first == fn) {
      if (e.second >= bytes) return HH_OK;
      HH_HIP(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
      e.second = bytes;
      return HH_OK;
    }
  HH_HIP(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  c->lds_attr.emplace_back(fn, bytes);
  return HH_OK;
}

struct ProfScope {  // hipEvent pair around one launch when profiling is on
  hh_ctx* c;
  EventPair* ep = nullptr;
  ProfScope(hh_ctx* ctx, int kind) : c(ctx) {
    if (!c->prof_now) return;
    if (c->events_used == c->events.size()) {
      EventPair p{};
      if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return;
      c->events.push_back(p);
    }
    ep = &c->events[c->events_used++];
    ep->kind = kind;
    (void)hipEventRecord(ep->a, c->stream);
  }
  ~ProfScope() {
    if (ep) (void)hipEventRecord(ep->b, c->stream);
  }
};

template <int N, int MODE>
int launch_first(hh_ctx* c, const FirstArgs& a, int batch) {
  using K = KA<N>;
  if (int rc = ensure_lds_attr(c, reinterpret_cast<const void*>(&k_first_pass<N, MODE>), (int)K::LDS)) return rc;
  ProfScope ps(c, 0);
  hipLaunchKernelGGL((k_first_pass<N, MODE>), dim3(K::NQ, batch + (a.fin.n > 0 ? 1 : 0)), dim3(K::THREADS), K::LDS,
                     c->stream, a);
  HH_HIP(c, hipGetLastError());
  return HH_OK;
}

template <int N, int EPI, int LOG>
int launch_second(hh_ctx* c, const SecondArgs& a, int batch) {
  using K = KB<N>;
  if (int rc = ensure_lds_attr(c, reinterpret_cast<const void*>(&k_second_pass<N, EPI, LOG>), (int)K::LDS)) return rc;
  ProfScope ps(c, 1);
  SecondArgs args = a;
  args.batch = batch;
  hipLaunchKernelGGL((k_second_pass<N, EPI, LOG>), dim3(a.n_kb, (batch + K::CPW - 1) / K::CPW), dim3(K::THREADS),
                     K::LDS, c->stream, args);
  HH_HIP(c, hipGetLastError());
  return HH_OK;
}

template <int MODE>
int dispatch_first(hh_ctx* c, const FirstArgs& a, int batch) {
  switch (c->n) {
    case 32: return launch_first<32, MODE>(c, a, batch);
    case 64: return launch_first<64, MODE>(c, a, batch);
    case 128: return launch_first<128, MODE>(c, a, batch);
    case 256: return launch_first<256, MODE>(c, a, batch);
    case 512: return launch_first<512, MODE>(c, a, batch);
    case 1024: return launch_first<1024, MODE>(c, a, batch);
  }
  return fail(c, HH_ERR_ARG, "unsupported image size");
}

template <int EPI, int LOG>
int dispatch_second_n(hh_ctx* c, const SecondArgs& a, int batch) {
  switch (c->n) {
    case 32: return launch_second<32, EPI, LOG>(c, a, batch);
    case 64: return launch_second<64, EPI, LOG>(c, a, batch);
    case 128: return launch_second<128, EPI, LOG>(c, a, batch);
    case 256: return launch_second<256, EPI, LOG>(c, a, batch);
    case 512: return launch_second<512, EPI, LOG>(c, a, batch);
    case 1024: return launch_second<1024, EPI, LOG>(c, a, batch);
  }
  return fail(c, HH_ERR_ARG, "unsupported image size");
}

template <int EPI>
int dispatch_second(hh_ctx* c, const SecondArgs& a, int batch) {
  if (EPI != EPI_STORE && a.log_flag) return dispatch_second_n<EPI, 1>(c, a, batch);
  return dispatch_second_n<EPI, 0>(c, a, batch);
}

// grow-only device buffer
int ensure_bytes(hh_ctx* c, void** p, size_t* cap, size_t need) {
  if (need <= *cap && *p) return HH_OK;
  if (*p) HH_HIP(c, hipFree(*p));
  *p = nullptr;
  *cap = 0;
  if (hipMalloc(p, need) != hipSuccess) {
    *p = nullptr;
    return fail(c, HH_ERR_NOMEM, "device allocation of " + std::to_string(need) + " bytes failed");
  }
  *cap = need;
  return HH_OK;
}

int ensure_img(hh_ctx* c, int64_t count) {
  const int64_t need = count * (int64_t)c->n * c->n;
  if (need <= c->cap_img) return HH_OK;
  if (c->d_img) HH_HIP(c, hipFree(c->d_img));
  c->d_img = nullptr;
  c->cap_img = 0;
  HH_HIP(c, hipMalloc(&c->d_img, need * sizeof(float)));
  c->cap_img = need;
  return HH_OK;
}

int ensure_spec(hh_ctx* c, int64_t count) {
  const int64_t need = count * (int64_t)(c->n / 2 + 1) * c->n;
  if (need <= c->cap_spec) return HH_OK;
  if (c->d_spec) HH_HIP(c, hipFree(c->d_spec));
  c->d_spec = nullptr;
  c->cap_spec = 0;
  HH_HIP(c, hipMalloc(&c->d_spec, need * sizeof(float2)));
  c->cap_spec = need;
  return HH_OK;
}

// images already in c->d_img ([count][N][N]); leaves the half-plane spectra in c->d_spec
int spectra_of_images(hh_ctx* c, int count) {
  int rc = ensure_spec(c, count);
  if (rc) return rc;
  for (int s0 = 0; s0 < count; s0 += c->max_batch) {
    const int nb = std::min(c->max_batch, count - s0);
    FirstArgs fa{};
    fa.images = c->d_img + (size_t)s0 * c->n * c->n;
    fa.twtab = c->d_tw;
    fa.inter = c->d_inter;
    fa.kb_mask = ~0ull;
    fa.g = c->geom;
    rc = dispatch_first<MODE_IMAGE>(c, fa, nb);
    if (rc) return rc;
    SecondArgs sa{};
    sa.inter = c->d_inter;
    sa.twtab = c->d_tw;
    sa.spec_out = c->d_spec + (size_t)s0 * (c->n / 2 + 1) * c->n;
    sa.n_kb = c->n / 16;
    rc = dispatch_second<EPI_STORE>(c, sa, nb);
    if (rc) return rc;
  }
  return HH_OK;
}

template <int N>
int launch_run_table(hh_ctx* c, const TableArgs& a, int runs, int rows) {
  using K = KT<N>;
  ProfScope ps(c, 3);
  hipLaunchKernelGGL((k_run_table<N>), dim3((rows + K::SUB - 1) / K::SUB, runs), dim3(256), 0, c->stream, a);
  HH_HIP(c, hipGetLastError());
  return HH_OK;
}

template <int N>
int launch_first_table(hh_ctx* c, const TableArgs& a, int batch) {
  using K = KT<N>;
  ProfScope ps(c, 0);
  const size_t lds = (size_t)a.rows_lds * K::KYW * sizeof(float2);
  hipLaunchKernelGGL((k_first_pass_table<N>), dim3(K::NQX * K::NQY, batch + (a.fin.n > 0 ? 1 : 0)), dim3(K::THREADS),
                     lds, c->stream, a);
  HH_HIP(c, hipGetLastError());
  return HH_OK;
}



int dispatch_run_table(hh_ctx* c, const TableArgs& a, int runs, int rows) {
#define HH_CALL(NN) launch_run_table<NN>(c, a, runs, rows)
  HH_SWITCH_N(c, HH_CALL);
#undef HH_CALL
}

int dispatch_first_table(hh_ctx* c, const TableArgs& a, int batch) {
#define HH_CALL(NN) launch_first_table<NN>(c, a, batch)
  HH_SWITCH_N(c, HH_CALL);
#undef HH_CALL
}

// A candidate list the shared-twist first pass can take: runs of `len` consecutive candidates with
// identical (twist, csym, rot) and positive finite rises; h_run_imax[r] = the subunit index range
// of run r (from its smallest rise).
struct RunPlan {
  bool ok = false;
  int64_t len = 0;
  int rows = 0;      // table rows per run: (2 max imax + 1) * n_units
  int rows_lds = 0;  // table rows a band of image columns can need (from the smallest rise); 0: too many
  bool fused = false;  // the fused pass fits: whole table slice + column factors in LDS
  int kg = 0;          // fused: table rows per group of four columns
  int rows_f = 0;      // fused: table rows staged per ky
};

constexpr int64_t HH_MIN_RUN = 8;              // shorter runs do not amortise their table and workgroup granularity
constexpr size_t HH_TABLE_BYTES_MAX = 1ull << 30;

size_t fused_lds(int n, int rows_lds, int kg) {
  switch (n) {
    case 32: return KF<32>::lds(rows_lds, kg);
    case 64: return KF<64>::lds(rows_lds, kg);
    case 128: return KF<128>::lds(rows_lds, kg);
    case 256: return KF<256>::lds(rows_lds, kg);
    case 512: return KF<512>::lds(rows_lds, kg);
    default: return KF<1024>::lds(rows_lds, kg);
  }
}

int table_cols(int n) {
  switch (n) {
    case 32: return KT<32>::COLS;
    case 64: return KT<64>::COLS;
    case 128: return KT<128>::COLS;
    case 256: return KT<256>::COLS;
    case 512: return KT<512>::COLS;
    default: return KT<1024>::COLS;
  }
}

RunPlan plan_runs(hh_ctx* c, const double* hp, int64_t g) {
  RunPlan plan;
  if (!hp || !c->table_path || c->geom.has_rot || g < HH_MIN_RUN) return plan;
  int64_t len = 1;
  while (len < g && hp[4 * len] == hp[0] && hp[4 * len + 2] == hp[2] && hp[4 * len + 3] == hp[3]) ++len;
  if (len < HH_MIN_RUN || g % len) return plan;
  const int64_t runs = g / len;
  c->h_run_imax.assign((size_t)runs, 0);
  int imax_all = 0;
  double rise_all = INFINITY;
  for (int64_t r = 0; r < runs; ++r) {
    const double* p = hp + 4 * r * len;
    double rise_min = INFINITY;
    for (int64_t k = 0; k < len; ++k) {
      const double* q = p + 4 * k;
      if (!(q[0] == p[0] && q[2] == p[2] && q[3] == p[3])) return plan;
      if (!(q[1] > 0.0) || !(q[1] < INFINITY)) return plan;
      rise_min = std::min(rise_min, q[1]);
    }
    const double im = std::ceil(c->geom.height / rise_min);
    if (im > 1048576.0) return plan;
    c->h_run_imax[(size_t)r] = (int)im;
    imax_all = std::max(imax_all, (int)im);
    rise_all = std::min(rise_all, rise_min);
  }
  // k_first_pass_table: rows = ceil(i1) - floor(i0) + 1 <= (i1 - i0) + 3 with
  // i1 - i0 = ((COLS - 1) apix + 2 rpx apix + 2 slack) / rise; one more row for float32 rounding
  const double span = ((double)(table_cols(c->n) - 1) + 2.0 * c->geom.rpx) * c->geom.apix + 2.0 * c->geom.slack;
  const double need = (std::floor(span / (double)(float)rise_all) + 4.0) * c->geom.n_units;
  plan.rows_lds = need <= 64.0 ? (int)need : 0;  // KT<N>::ROWS_MAX: one lane per staged row
  plan.rows = (2 * imax_all + 1) * c->geom.n_units;
  if ((size_t)plan.rows * (c->n / 2) * sizeof(float2) > HH_TABLE_BYTES_MAX) return plan;
  // fused pass: subunit indices i with lo <= i rise <= hi for four columns, hi - lo = (3 + 2 rpx) apix + 2 slack
  if (c->fused_path) {
    const double span4 = (3.0 + 2.0 * c->geom.rpx) * c->geom.apix + 2.0 * c->geom.slack;
    const double kg = (std::floor(span4 / (double)(float)rise_all) + 2.0) * c->geom.n_units;
    int rows_f = std::max(plan.rows, (int)std::min(kg, 1.0e6));
    rows_f += (4 - rows_f % 8 + 8) % 8;  // = 4 (mod 8): the eight ky rows of the slice start in different banks
    if (kg <= 16.0 && fused_lds(c->n, rows_f, (int)kg) <= 160 * 1024 - 1024) {
      plan.fused = true;
      plan.kg = (int)kg;
      plan.rows_f = rows_f;
    }
  }
  if (!plan.fused && plan.rows_lds == 0) return plan;
  plan.len = len;
  plan.ok = true;
  return plan;
}

// Second pass + scores of one batch whose intermediate is in c->d_inter.
// Scores of one batch whose moments (and, with several segments, q) the second pass has left.
// Second pass + scores of one batch whose intermediate is in c->d_inter.
// Scores of one batch whose moments (and, with several segments, q) the second pass has left.
int scores_tail(hh_ctx* c, int64_t g, int64_t g0, int nb, bool last, float* d_scores, FinArgs& pending,
                const double* partials = nullptr) {
  const int npart = npart_for(c->n);
  if (!partials) partials = c->d_partials;
  if (c->n_segments == 1) {
    float* const out = d_scores + g0;
    if (!last) {  // the next launch of this range scores the batch in its grid layer 0
      pending = FinArgs{partials, out, nb, npart, c->ref[0]};
    } else {
      ProfScope ps(c, 2);
      const int threads = c->n < 64 ? c->n : 256;  // teams of min(64, n) lanes, one candidate each
      hipLaunchKernelGGL(k_finalize, dim3(std::min(1024, (nb + 3) / 4)), dim3(threads), 0, c->stream, partials,
                         npart, (int64_t)nb, c->ref[0], out);
    }
    HH_HIP(c, hipGetLastError());
  } else {
    // several segments: one MFMA contraction of the batch's q against all segments' centred
    // spectra, then Pearson per (segment, candidate)
    // compact q: the flat axis of weighted bins in slices of 512; else one spectrum row per slice, those the mask gives any weight
    const int rows = c->q_slices ? c->q_slices : c->n_seg_rows;
    const size_t K = c->q_slices ? c->q_stride : (size_t)(c->n / 2 + 1) * c->n;
    ProfScope ps(c, 2);
    hipLaunchKernelGGL(k_segment_corr, dim3(rows, (nb + 255) / 256, c->s_pad / 64), dim3(256), 0, c->stream, c->d_q,
                       c->d_wec, c->q_slices ? 512 : c->n, K, c->b_pad, c->s_pad, c->d_cpart, c->q_slices ? (const int*)nullptr : c->d_seg_rows);
    const int total = nb * c->n_segments;
    hipLaunchKernelGGL(k_sum_partials, dim3(std::min(1024, (nb + 3) / 4)), dim3(256), 0, c->stream, partials, npart,
                       nb, c->d_psum);
    hipLaunchKernelGGL(k_finalize_segments, dim3((total + 255) / 256), dim3(256), 0, c->stream, c->d_psum, 1,
                       c->d_cpart, rows, c->b_pad, c->s_pad, nb, c->n_segments, c->d_ref, d_scores, g, g0);
    HH_HIP(c, hipGetLastError());
  }
  return HH_OK;
}

// Second pass + scores of one batch whose intermediate is in c->d_inter.
int second_and_scores(hh_ctx* c, int64_t g, int64_t g0, int nb, bool last, float* d_scores, FinArgs& pending) {
  SecondArgs sa{};
  sa.inter = c->d_inter;
  sa.twtab = c->d_tw;
  sa.w2 = c->d_w2;
  sa.partials = c->d_partials;
  sa.log_flag = c->log_flag;
  sa.kb_list = c->d_kb_list;
  sa.n_kb = c->n_kb;
  int rc;
  if (c->n_segments == 1) {
    rc = dispatch_second<EPI_SCORE>(c, sa, nb);
  } else {
    sa.q_out = c->d_q;  // several segments: q of the batch -> HBM for the contraction
    sa.q_roff = c->q_slices ? c->d_q_roff : nullptr;
    sa.q_stride = c->q_stride;
    rc = dispatch_second<EPI_QSTORE>(c, sa, nb);
  }
  if (rc) return rc;
  return scores_tail(c, g, g0, nb, last, d_scores, pending);
}

template <int N>
int launch_factors(hh_ctx* c, const FactorArgs& a, int batch) {
  ProfScope ps(c, 0);  // reported in the first-pass slot: it is what is left of the first pass
  FactorArgs args = a;
  args.count = batch;
  hipLaunchKernelGGL((k_column_factors<N>), dim3(batch), dim3(N), (size_t)(a.rows_lds + 1) * sizeof(float), c->stream, args);
  HH_HIP(c, hipGetLastError());
  return HH_OK;
}

template <int N, int EPI, int LOG>
int launch_fused(hh_ctx* c, const FusedArgs& a, int layers) {
  using K = KF<N>;
  if (int rc = ensure_lds_attr(c, reinterpret_cast<const void*>(&k_fused_pass<N, EPI, LOG>), 160 * 1024)) return rc;
  ProfScope ps(c, 1);
  hipLaunchKernelGGL((k_fused_pass<N, EPI, LOG>), dim3(a.n_kb, a.factor_layers + layers + (a.fin.n > 0 ? 1 : 0)),
                     dim3(K::THREADS), K::lds(a.rows_lds, a.kg), c->stream, a);
  HH_HIP(c, hipGetLastError());
  return HH_OK;
}

int dispatch_factors(hh_ctx* c, const FactorArgs& a, int batch) {
#define HH_CALL(NN) launch_factors<NN>(c, a, batch)
  HH_SWITCH_N(c, HH_CALL);
#undef HH_CALL
}

template <int EPI, int LOG>
int dispatch_fused_n(hh_ctx* c, const FusedArgs& a, int layers) {
#define HH_CALL(NN) launch_fused<NN, EPI, LOG>(c, a, layers)
  HH_SWITCH_N(c, HH_CALL);
#undef HH_CALL
}

// Workgroups of k_fused_pass the device holds at once (occupancy x compute units), for the launch shape in `a`.
template <int N, int EPI, int LOG>
int slots_fused(hh_ctx* c, const FusedArgs& a, int* out) {
  using K = KF<N>;
  if (int rc = ensure_lds_attr(c, reinterpret_cast<const void*>(&k_fused_pass<N, EPI, LOG>), 160 * 1024)) return rc;
  int per_cu = 0;
  HH_HIP(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_fused_pass<N, EPI, LOG>, K::THREADS, K::lds(a.rows_lds, a.kg)));
  *out = std::max(1, per_cu) * std::max(1, c->n_cu);
  return HH_OK;
}
template <int EPI, int LOG>
int slots_fused_n(hh_ctx* c, const FusedArgs& a, int* out) {
#define HH_CALL(NN) slots_fused<NN, EPI, LOG>(c, a, out)
  HH_SWITCH_N(c, HH_CALL);
#undef HH_CALL
}
int fused_slots(hh_ctx* c, const FusedArgs& a, int* out) {
  const size_t key = ((size_t)a.rows_lds << 20) ^ ((size_t)a.kg << 8) ^ (size_t)(c->n_segments > 1) ^ ((size_t)c->log_flag << 1);
  if (c->slots_key == key && c->slots > 0) { *out = c->slots; return HH_OK; }
  int rc;
  if (c->n_segments == 1) rc = c->log_flag ? slots_fused_n<EPI_SCORE, 1>(c, a, out) : slots_fused_n<EPI_SCORE, 0>(c, a, out);
  else rc = c->log_flag ? slots_fused_n<EPI_QSTORE, 1>(c, a, out) : slots_fused_n<EPI_QSTORE, 0>(c, a, out);
  if (rc == HH_OK) { c->slots = *out; c->slots_key = key; }
  return rc;
}

// How the runs of a launch are cut into workgroups.  The grid is (layers of one run) x ky blocks workgroups, `slots`
// of them resident at a time, each paying a fixed set-up (table slice, weights, twiddles: about four candidates'
// worth) before its candidates; a launch lasts about ceil(workgroups / slots) x (candidates per workgroup + set-up).
// Few, long workgroups amortise the set-up, but the last, partly filled round of a launch costs a whole one.  So the
// runs that fill whole rounds get `groups_a` layers each, and the runs left over for the last round are cut finer
// (`groups_b`), so that round is short.  C2 in one launch (400 runs of 250, 32 ky blocks, 512 slots) is exactly 25
// rounds of whole runs; an eighth of it (50 runs) is 3 rounds of whole runs + 2 runs in 8 pieces each, 798 units
// instead of 4 x 254.
struct FusedSchedule { int runs_a, groups_a, cpw_a, groups_b, cpw_b, layers; };
FusedSchedule fused_schedule(int64_t runs, int run_len, int n_kb, int slots) {
  const int setup = 4, min_cpw = 16, gmax = std::max(1, run_len / min_cpw);
  auto cpw_of = [&](int g) { return (run_len + g - 1) / g; };
  auto groups_of = [&](int cpw) { return (run_len + cpw - 1) / cpw; };   // even groups: ceil(len / ceil(len / g)) <= g
  FusedSchedule best{};
  double best_t = 1e300;
  const int g_lo = HH_KF_CPW > 0 ? groups_of(HH_KF_CPW) : 1, g_hi = HH_KF_CPW > 0 ? g_lo : gmax;
  for (int ga = g_lo; ga <= g_hi; ++ga) {
    const int cpw_a = cpw_of(ga), ga_e = groups_of(cpw_a);
    const int64_t per_run = (int64_t)ga_e * n_kb;
    const int64_t full = runs * per_run / slots;                          // whole rounds of region A
    const int64_t runs_a = HH_KF_CPW > 0 ? runs : std::min<int64_t>(runs, full * slots / per_run);
    const int64_t runs_b = runs - runs_a;
    const double ta = (double)((runs_a * per_run + slots - 1) / slots) * (cpw_a + setup);
    int gb = ga_e, cpw_b = cpw_a;
    double tb = 0;
    if (runs_b > 0) {
      tb = 1e300;
      for (int g = ga; g <= gmax; ++g) {
        const int cw = cpw_of(g), ge = groups_of(cw);
        const double t = (double)((runs_b * ge * n_kb + slots - 1) / slots) * (cw + setup);
        if (t < tb * 0.995) { tb = t; gb = ge; cpw_b = cw; }
      }
    }
    if (ta + tb < best_t * 0.995) {  // (near-ties go to the fewer, longer workgroups)
      best_t = ta + tb;
      best = FusedSchedule{(int)runs_a, ga_e, cpw_a, gb, cpw_b, (int)(runs_a * ga_e + runs_b * gb)};
    }
  }
  return best;
}

int dispatch_fused(hh_ctx* c, const FusedArgs& a, int layers) {
  if (c->n_segments == 1) return c->log_flag ? dispatch_fused_n<EPI_SCORE, 1>(c, a, layers) : dispatch_fused_n<EPI_SCORE, 0>(c, a, layers);
  return c->log_flag ? dispatch_fused_n<EPI_QSTORE, 1>(c, a, layers) : dispatch_fused_n<EPI_QSTORE, 0>(c, a, layers);
}

// The sweep with the shared-twist first pass (plan.ok): batches are whole runs (or pieces of one
// run); the tables of as many runs as fit HH_TABLE_BYTES_MAX are built by one launch ahead of them.
// d_partials holds two halves of `batch` candidates each (zero-filled: rows of ky blocks the mask skips
// are never written)
int ensure_partials(hh_ctx* c, int batch) {
  if (batch <= c->cap_partials) return HH_OK;
  if (c->d_partials) HH_HIP(c, hipFree(c->d_partials));
  c->d_partials = nullptr;
  c->cap_partials = 0;
  const size_t bytes = (size_t)2 * batch * npart_for(c->n) * 3 * sizeof(double);
  HH_HIP(c, hipMalloc(&c->d_partials, bytes));
  HH_HIP(c, hipMemsetAsync(c->d_partials, 0, bytes, c->stream));
  c->cap_partials = batch;
  return HH_OK;
}

// several segments: the per-batch buffers (masked q [b_pad][K], covariance numerators [N/2+1][b_pad][s_pad], summed
// moments [b_pad][3]) for batches of up to `batch` candidates.  They only grow, and they are sized by what a sweep really
// launches — round 2 sized them at hh_set_reference for the largest batch any sweep could use (12.9 GB of q at N = 512 for
// a 2-segment, 1-candidate call) and cleared all of it on every call.
int ensure_segment_buffers(hh_ctx* c, int batch) {
  const int want = (std::max(batch, 64) + 63) / 64 * 64;
  if (want <= c->b_pad && c->d_q && c->d_cpart && c->d_psum) return HH_OK;
  const size_t nh = c->q_stride ? c->q_stride : (size_t)(c->n / 2 + 1) * c->n;
  int rc;
  if ((rc = ensure_bytes(c, (void**)&c->d_q, &c->cap_q, (size_t)want * nh * sizeof(float)))) return rc;
  if ((rc = ensure_bytes(c, (void**)&c->d_cpart, &c->cap_cpart, (size_t)(c->n / 2 + 1) * want * c->s_pad * sizeof(float)))) return rc;
  if ((rc = ensure_bytes(c, (void**)&c->d_psum, &c->cap_psum, (size_t)want * 3 * sizeof(double)))) return rc;
  HH_HIP(c, hipMemsetAsync(c->d_q, 0, (size_t)want * nh * sizeof(float), c->stream));  // pad rows of a batch (and, compact, the pad bins of a row of q) stay finite
  c->b_pad = want;
  return HH_OK;
}

// several segments: candidates per launch of the shared-twist pipelines (q of a batch lives in HBM)
inline int seg_batch(int n) {
  const int64_t per = (int64_t)(n / 2 + 1) * n * (int64_t)sizeof(float);
  return (int)std::max<int64_t>(1024, std::min<int64_t>(HH_SEG_BATCH, HH_SEG_BYTES / per) / 64 * 64);
}

// Candidates [first, first + count) of the list of g (count = whole runs of plan.len).
int sweep_runs(hh_ctx* c, const double* d_params, int64_t g, float* d_scores, const RunPlan& plan, int64_t first_cand,
               int64_t count_cand) {
  d_params += 4 * first_cand;  // indices below are relative to the range; scores and g0 get first_cand back
  const int64_t runs = count_cand / plan.len;
  const int nky = c->n / 2;
  // The fused pass has no intermediate to hold, so its batches are not tied to max_batch: long
  // launches even out the tail of the grid (C2: 3.2 M candidates/s at 250 per launch, 3.8 M at 4000).
  const int fused_cap = (int)std::max<int64_t>(1024, std::min<int64_t>(HH_FUSED_BATCH, HH_FUSED_BYTES / ((int64_t)2 * std::max(1, plan.kg) * c->n * 4)));
  // (never more than the sweep itself holds: the factor and moment buffers are sized by it and only grow)
  const int bmax = !plan.fused ? c->max_batch
                   : (int)std::min<int64_t>(c->n_segments == 1 ? std::max(c->max_batch, fused_cap) : std::max(c->max_batch, seg_batch(c->n)),
                                            std::max<int64_t>(count_cand, 16));
  if (c->n_segments > 1) {
    const int rcs = ensure_segment_buffers(c, bmax);
    if (rcs) return rcs;
  }
  const int64_t per_batch = plan.len <= bmax ? bmax / plan.len : 1;
  const size_t run_bytes = (size_t)plan.rows * nky * sizeof(float2);
  int64_t per_group = std::max<int64_t>(1, (int64_t)(HH_TABLE_BYTES_MAX / run_bytes));
  per_group = std::min<int64_t>(std::min<int64_t>(per_group, runs), 65535);
  if (per_group > per_batch) per_group -= per_group % per_batch;
  const size_t need = (size_t)per_group * run_bytes;
  if (need > c->cap_table) {
    if (c->d_table) HH_HIP(c, hipFree(c->d_table));
    c->d_table = nullptr;
    c->cap_table = 0;
    HH_HIP(c, hipMalloc(&c->d_table, need));
    c->cap_table = need;
  }
  if (plan.fused) {
    const size_t need_eg = (size_t)2 * bmax * plan.kg * c->n * sizeof(float);  // two halves, alternating
    if (need_eg > c->cap_eg) {
      if (c->d_eg) HH_HIP(c, hipFree(c->d_eg));
      c->d_eg = nullptr;
      c->cap_eg = 0;
      HH_HIP(c, hipMalloc(&c->d_eg, need_eg));
      c->cap_eg = need_eg;
    }
    if ((size_t)bmax > c->cap_cgs) {
      if (c->d_cgs) HH_HIP(c, hipFree(c->d_cgs));
      c->d_cgs = nullptr;
      c->cap_cgs = 0;
      HH_HIP(c, hipMalloc(&c->d_cgs, (size_t)2 * bmax * (c->n / 4 + 4) * sizeof(int)));
      c->cap_cgs = (size_t)bmax;
    }
    const int rc = ensure_partials(c, bmax);
    if (rc) return rc;
  }
  if (runs > c->cap_runs) {
    if (c->d_run_imax) HH_HIP(c, hipFree(c->d_run_imax));
    c->d_run_imax = nullptr;
    c->cap_runs = 0;
    HH_HIP(c, hipMalloc(&c->d_run_imax, (size_t)runs * sizeof(int)));
    c->cap_runs = runs;
  }
  HH_HIP(c, hipMemcpyAsync(c->d_run_imax, c->h_run_imax.data(), (size_t)runs * sizeof(int), hipMemcpyHostToDevice,
                           c->stream));
  // the batches of the sweep: whole runs (or pieces of one run), in order
  struct Batch {
    int64_t g0;      // first candidate
    int nb;          // candidates
    int64_t r0;      // first run
    int run_len;     // candidates per run inside the batch (a piece of one run counts as one run)
    int64_t q0;      // first run of the table group the batch belongs to
  };
  std::vector<Batch> batches;
  for (int64_t q0 = 0; q0 < runs; q0 += per_group) {
    const int nq = (int)std::min<int64_t>(per_group, runs - q0);
    for (int64_t r0 = q0; r0 < q0 + nq; r0 += per_batch) {
      const int nr = (int)std::min<int64_t>(per_batch, q0 + nq - r0);
      const int64_t first = r0 * plan.len, count = (int64_t)nr * plan.len;
      for (int64_t g0 = first; g0 < first + count; g0 += bmax) {
        const int nb = (int)std::min<int64_t>(bmax, first + count - g0);
        batches.push_back(Batch{g0, nb, r0, (int)std::min<int64_t>(plan.len, nb), q0});
      }
    }
  }
  const size_t eg_half = (size_t)bmax * plan.kg * c->n;
  const size_t cg_half = (size_t)bmax * (c->n / 4 + 4);  // cgs_stride<N>()
  auto factor_args = [&](const Batch& bt, int half) {
    FactorArgs fa{};
    fa.params = d_params + 4 * bt.g0;
    fa.units = c->d_units;
    fa.run_imax = c->d_run_imax + bt.r0;
    fa.eg = c->d_eg + half * eg_half;
    fa.cgs = c->d_cgs + half * cg_half;
    fa.run_len = bt.run_len;
    fa.kg = plan.kg;
    fa.rows_lds = plan.rows_f;
    fa.count = bt.nb;
    fa.g = c->geom;
    return fa;
  };

  TableArgs ta{};
  ta.units = c->d_units;
  ta.twtab = c->d_tw;
  ta.inter = c->d_inter;
  ta.kb_mask = c->kb_mask;
  ta.cap = plan.rows;
  ta.rows_lds = plan.rows_lds;
  ta.g = c->geom;
  FinArgs pending{};
  int64_t table_group = -1;
  for (size_t bi = 0; bi < batches.size(); ++bi) {
    const Batch& bt = batches[bi];
    if (bt.q0 != table_group) {  // this group's tables, one launch
      table_group = bt.q0;
      const int nq = (int)std::min<int64_t>(per_group, runs - bt.q0);
      c->prof_now = c->profiling > 0;
      ta.params = d_params + 4 * bt.q0 * plan.len;
      ta.table = c->d_table;
      ta.run_imax = c->d_run_imax + bt.q0;
      ta.run_len = (int)std::min<int64_t>(plan.len, 1 << 30);
      ta.fin = FinArgs{};
      int rows = 0;
      for (int r = 0; r < nq; ++r) rows = std::max(rows, (2 * c->h_run_imax[(size_t)(bt.q0 + r)] + 1) * c->geom.n_units);
      const int rc = dispatch_run_table(c, ta, nq, rows);
      if (rc) return rc;
    }
    c->prof_now = c->profiling > 0 && ((int64_t)bi % c->profiling) == 0;
    if (c->prof_now) c->prof_candidates += bt.nb;
    ta.table = c->d_table + (size_t)(bt.r0 - bt.q0) * plan.rows * nky;
    ta.run_imax = c->d_run_imax + bt.r0;
    ta.run_len = plan.len <= bmax ? (int)plan.len : bmax + 1;
    ta.params = d_params + 4 * bt.g0;
    int rc;
    if (plan.fused) {
      // build + transform + moments without an intermediate.  The batch's column factors were
      // computed by leading layers of the previous batch's launch (its own launch for the first).
      const int half = (int)(bi & 1);
      if (bi == 0) {
        rc = dispatch_factors(c, factor_args(bt, half), bt.nb);
        if (rc) return rc;
      }
      FusedArgs fu{};
      fu.params = ta.params;
      fu.twtab = c->d_tw;
      fu.table = ta.table;
      fu.run_imax = ta.run_imax;
      fu.eg = c->d_eg + half * eg_half;
      fu.cgs = c->d_cgs + half * cg_half;
      fu.w2 = c->d_w2;
      double* const part = c->d_partials + (size_t)half * bmax * npart_for(c->n) * 3;
      fu.partials = part;
      fu.q_out = c->n_segments > 1 ? c->d_q : nullptr;
      fu.q_roff = c->n_segments > 1 && c->q_slices ? c->d_q_roff : nullptr;
      fu.q_stride = c->q_stride;
      fu.kb_list = c->d_kb_list;
      fu.n_kb = c->n_kb;
      fu.batch = bt.nb;
      fu.run_len = bt.run_len;
      fu.cap = plan.rows;
      fu.rows_lds = plan.rows_f;
      fu.kg = plan.kg;
      int work_layers = 0;
      {
        const int64_t runs_b = (bt.nb + fu.run_len - 1) / fu.run_len;
        const int len = std::min(fu.run_len, bt.nb);
        int slots = 0;
        rc = fused_slots(c, fu, &slots);
        if (rc) return rc;
        const FusedSchedule fs = fused_schedule(runs_b, len, c->n_kb, slots);
        fu.runs_a = fs.runs_a; fu.groups_a = fs.groups_a; fu.cpw_a = fs.cpw_a; fu.groups_b = fs.groups_b; fu.cpw_b = fs.cpw_b;
        work_layers = fs.layers;
      }
      fu.n_units = c->geom.n_units;
      fu.fin = pending;
      if (bi + 1 < batches.size()) {
        fu.next = factor_args(batches[bi + 1], half ^ 1);
        fu.factor_layers = (fu.next.count + c->n_kb - 1) / c->n_kb;
      }
      rc = dispatch_fused(c, fu, work_layers);
      if (rc) return rc;
      pending = FinArgs{};
      rc = scores_tail(c, g, first_cand + bt.g0, bt.nb, bi + 1 == batches.size(), d_scores, pending, part);
      if (rc) return rc;
      continue;
    }
    ta.fin = pending;
    rc = dispatch_first_table(c, ta, bt.nb);
    if (rc) return rc;
    pending = FinArgs{};
    rc = second_and_scores(c, g, first_cand + bt.g0, bt.nb, bi + 1 == batches.size(), d_scores, pending);
    if (rc) return rc;
  }
  c->prof_now = false;
  return HH_OK;
}

// Candidates [first, first + count) through the general pipeline (raster + two transforms each).
int sweep_transform(hh_ctx* c, const double* d_params, int64_t g, float* d_scores, int64_t first, int64_t count) {
  FinArgs pending{};
  int64_t batch_no = 0;
  for (int64_t g0 = first; g0 < first + count; g0 += c->max_batch, ++batch_no) {
    const int nb = (int)std::min<int64_t>(c->max_batch, first + count - g0);
    c->prof_now = c->profiling > 0 && (batch_no % c->profiling) == 0;
    if (c->prof_now) c->prof_candidates += nb;
    FirstArgs fa{};
    fa.params = d_params + 4 * g0;
    fa.units = c->d_units;
    fa.twtab = c->d_tw;
    fa.inter = c->d_inter;
    fa.fin = pending;
    fa.kb_mask = c->kb_mask;
    fa.g = c->geom;
    int rc = dispatch_first<MODE_RASTER>(c, fa, nb);
    if (rc) return rc;
    pending = FinArgs{};
    rc = second_and_scores(c, g, g0, nb, g0 + nb == first + count, d_scores, pending);
    if (rc) return rc;
  }
  c->prof_now = false;
  return HH_OK;
}

// ld: row stride of d_scores (segment s of candidate i goes to d_scores[s * ld + i]); the helpers below only use
// their `g` argument for that addressing, so they are handed ld.
int sweep_on_device(hh_ctx* c, const double* d_params, const int64_t n_cand, float* d_scores,
                    const double* h_params = nullptr, int64_t ld = 0) {
  if (c->general) return gen_sweep(c, d_params, h_params, n_cand, d_scores, ld);
  RunPlan plan = plan_runs(c, h_params, n_cand);
  c->last_first_pass = plan.ok ? (plan.fused ? 2 : 1) : 0;
  const int64_t g = ld > 0 ? ld : n_cand;
  if (plan.ok) return sweep_runs(c, d_params, g, d_scores, plan, 0, n_cand);
  // A list that starts inside a run (a shard cut anywhere) or ends with a partial run: the whole runs in
  // the middle still take the shared-twist pipeline, the two ragged ends the general one.
  if (h_params && c->table_path && !c->geom.has_rot && n_cand >= 2 * HH_MIN_RUN) {
    auto same = [&](int64_t i, int64_t j) {
      return h_params[4 * i] == h_params[4 * j] && h_params[4 * i + 2] == h_params[4 * j + 2] &&
             h_params[4 * i + 3] == h_params[4 * j + 3];
    };
    int64_t head = 1;
    while (head < n_cand && same(head, 0)) ++head;
    int64_t len = 1;
    while (head + len < n_cand && same(head + len, head)) ++len;
    const int64_t count = head < n_cand ? (n_cand - head) / len * len : 0;
    if (head < n_cand && head < len && len >= HH_MIN_RUN && count >= len) {
      plan = plan_runs(c, h_params + 4 * head, count);
      if (plan.ok) {
        c->last_first_pass = plan.fused ? 2 : 1;
        int rc = sweep_transform(c, d_params, g, d_scores, 0, head);
        if (rc) return rc;
        rc = sweep_runs(c, d_params, g, d_scores, plan, head, count);
        if (rc) return rc;
        return head + count < n_cand ? sweep_transform(c, d_params, g, d_scores, head + count, n_cand - head - count) : HH_OK;
      }
    }
    // aligned start, partial last run
    len = head;
    const int64_t whole = n_cand / len * len;
    if (len >= HH_MIN_RUN && whole >= len && whole < n_cand) {
      plan = plan_runs(c, h_params, whole);
      if (plan.ok) {
        c->last_first_pass = plan.fused ? 2 : 1;
        const int rc = sweep_runs(c, d_params, g, d_scores, plan, 0, whole);
        if (rc) return rc;
        return sweep_transform(c, d_params, g, d_scores, whole, n_cand - whole);
      }
    }
  }
  return sweep_transform(c, d_params, g, d_scores, 0, n_cand);
}

int check_ready(hh_ctx* c, bool need_ref) {
  if (!c) return HH_ERR_ARG;
  if (!c->have_geom) return fail(c, HH_ERR_STATE, "hh_set_geometry has not been called");
  if (need_ref && c->n_segments == 0) return fail(c, HH_ERR_STATE, "hh_set_reference has not been called");
  return HH_OK;
}

template <int N>
int launch_filter(hh_ctx* c, float f2_lp, float f2_hp, const float2* spec, float2* rows, float* image) {
  hipLaunchKernelGGL((k_filter_rows<N>), dim3(N / 2 + 1), dim3(N / 8), 0, c->stream, spec, c->d_tw, f2_lp, f2_hp, rows);
  hipLaunchKernelGGL((k_inverse_cols<N>), dim3(N / 2), dim3(N / 8), 0, c->stream, rows, c->d_tw, image);
  HH_HIP(c, hipGetLastError());
  return HH_OK;
}

template <typename T>
int pair_reduce(hh_ctx* c, const T* a, const T* b, int64_t n, double mx, double my, double out[5]) {
  const int grid = (int)std::min<int64_t>(1024, (n + 255) / 256);
  T *da = nullptr, *db = nullptr;
  double* dp = nullptr;
  std::vector<double> hp((size_t)grid * 5);
  hipError_t e = hipMalloc(&da, n * sizeof(T));
  if (e == hipSuccess) e = hipMalloc(&db, n * sizeof(T));
  if (e == hipSuccess) e = hipMalloc(&dp, (size_t)grid * 5 * sizeof(double));
  if (e == hipSuccess) e = hipMemcpyAsync(da, a, n * sizeof(T), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(db, b, n * sizeof(T), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL((k_pair_sums<T>), dim3(grid), dim3(256), 0, c->stream, da, db, n, mx, my, dp);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(hp.data(), dp, hp.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(da);
  (void)hipFree(db);
  (void)hipFree(dp);
  if (e != hipSuccess) return fail(c, HH_ERR_HIP, std::string("pair_reduce: ") + hipGetErrorString(e));
  for (int k = 0; k < 5; ++k) out[k] = 0;
  for (int g = 0; g < grid; ++g)
    for (int k = 0; k < 5; ++k) out[k] += hp[(size_t)g * 5 + k];
  return HH_OK;
}

// Two passes like analysis.py:793-799: means first, then the centred sums.
template <typename T>
int pearson(hh_ctx* c, const T* a, const T* b, int64_t n, double* out) {
  if (!c || !a || !b || !out || n <= 0) return fail(c, HH_ERR_ARG, "cross_correlation: bad argument");
  HH_HIP(c, hipSetDevice(c->device));
  double s[5];
  int rc = pair_reduce(c, a, b, n, 0.0, 0.0, s);
  if (rc) return rc;
  const double mx = s[0] / (double)n, my = s[1] / (double)n;
  rc = pair_reduce(c, a, b, n, mx, my, s);
  if (rc) return rc;
  const double norm = std::sqrt(s[2] * s[3]);
  *out = norm == 0 ? 0.0 : s[4] / norm;
  return HH_OK;
}

template <typename T>
int cosine(hh_ctx* c, const T* a, const T* b, int64_t n, double* out) {
  if (!c || !a || !b || !out || n <= 0) return fail(c, HH_ERR_ARG, "cosine_similarity: bad argument");
  HH_HIP(c, hipSetDevice(c->device));
  double s[5];
  int rc = pair_reduce(c, a, b, n, 0.0, 0.0, s);
  if (rc) return rc;
  const double norm = std::sqrt(s[2]) * std::sqrt(s[3]);
  *out = norm == 0 ? 0.0 : s[4] / norm;
  return HH_OK;
}

#include "general_host.inc"  // host side of the general-size path

}  // namespace

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

int hh_abi_version(void) { return HH_ABI_VERSION; }

int hh_device_count(int* count) try {
  if (!count) return HH_ERR_ARG;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    *count = 0;
    return fail(nullptr, HH_ERR_HIP, "hipGetDeviceCount failed");
  }
  *count = n;
  return HH_OK;
} HH_CATCH_CTX(nullptr, "hh_device_count")

int hh_selftest_exception(int kind) try {
  if (kind == 0) {
    std::vector<double> v;
    v.resize(v.max_size() + (size_t)1);
    return (int)v.size();
  }
  if (kind == 1) {
    volatile size_t absurd = (size_t)1 << 62;
    std::vector<char> v;
    v.reserve(absurd);   // below max_size(): the allocator itself refuses
    return (int)v.capacity();
  }
  if (kind == 2) throw std::system_error(std::make_error_code(std::errc::resource_unavailable_try_again), "thread");
  if (kind == 3) throw 42;
  return HH_OK;
} HH_CATCH_CTX(nullptr, "hh_selftest_exception")

int64_t hh_algorithmic_bytes(int n) { return 4LL * n * n + 16LL * n * (n / 2 + 1); }

const char* hh_last_error(const hh_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int hh_create(hh_ctx** out, int device, int n, int max_batch) { return hh_create2(out, device, n, n, max_batch); }

int hh_create2(hh_ctx** out, int device, int ny, int nx, int max_batch) try {
  if (!out) return fail(nullptr, HH_ERR_ARG, "hh_create: out is NULL");
  *out = nullptr;
  const bool general = !(ny == nx && supported_n(ny));
  const int n = general ? 0 : ny;
  if (general && (ny < 8 || nx < 8 || ny > 1024 || nx > 1024))
    return fail(nullptr, HH_ERR_ARG, "hh_create: image sides must lie in [8, 1024]");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, HH_ERR_HIP, "hh_create: no HIP device is visible (the sweep has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(nullptr, HH_ERR_ARG, "hh_create: device ordinal out of range");
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess)
    return fail(nullptr, HH_ERR_HIP, "hh_create: hipGetDeviceProperties failed");
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, HH_ERR_HIP, std::string("hh_create: built for gfx950 only, device is ") + prop.gcnArchName);
  if (max_batch <= 0 && !general) {
    // one batch of half spectra (N^2 * 4 B each) = 256 MiB, the size of the Infinity Cache:
    // measured best on MI355X (larger batches fall out of the cache, smaller ones leave CUs idle
    // at the tails of the two kernels)
    max_batch = (int)std::max<int64_t>(16, (256LL << 20) / ((int64_t)n * n * 4));
    max_batch = std::min(max_batch, 4096);
  }
  if (max_batch > 65534) max_batch = 65534;  // gridDim.y, plus the finalize layer of k_first_pass
  hh_ctx* c = new (std::nothrow) hh_ctx();
  if (!c) return fail(nullptr, HH_ERR_NOMEM, "hh_create: out of host memory");
  c->device = device;
  c->n = n;
  c->ny = ny;
  c->nx = nx;
  c->general = general;
  c->max_batch = max_batch;
  struct Guard { hh_ctx* c; ~Guard() { if (c) hh_destroy(c); } } guard{c};   // an exception on the way releases the half-built context
  auto bail = [&](int code, const std::string& msg) {
    g_create_error = msg;
    guard.c = nullptr;
    hh_destroy(c);
    return code;
  };
#define HH_CREATE_HIP(call)                                                                     \
  do {                                                                                          \
    hipError_t e__ = (call);                                                                    \
    if (e__ != hipSuccess) return bail(HH_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
  } while (0)
  HH_CREATE_HIP(hipSetDevice(device));
  HH_CREATE_HIP(hipDeviceGetAttribute(&c->n_cu, hipDeviceAttributeMultiprocessorCount, device));
  HH_CREATE_HIP(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
  c->stream = c->own_stream;
  if (general) {  // runtime-sized kernels; buffers grow with the sweeps
    HH_CREATE_HIP(hipMalloc(&c->d_units, (size_t)HH_MAX_UNITS * 3 * sizeof(double)));
    c->max_batch = (int)GEN_BATCH;
    const int rcg = gen_create(c, ny, nx);
    if (rcg) return bail(rcg, c->err);
    guard.c = nullptr;
    *out = c;
    return HH_OK;
  }
  HH_CREATE_HIP(hipMalloc(&c->d_tw, (size_t)n * sizeof(float2)));
  HH_CREATE_HIP(hipMalloc(&c->d_inter, (size_t)max_batch * (n / 2) * n * sizeof(float2)));
  HH_CREATE_HIP(hipMalloc(&c->d_partials, (size_t)2 * max_batch * npart_for(n) * 3 * sizeof(double)));
  HH_CREATE_HIP(hipMemset(c->d_partials, 0, (size_t)2 * max_batch * npart_for(n) * 3 * sizeof(double)));
  c->cap_partials = max_batch;
  HH_CREATE_HIP(hipMalloc(&c->d_units, (size_t)HH_MAX_UNITS * 3 * sizeof(double)));

  std::vector<float2> tw((size_t)n);
  for (int k = 0; k < n; ++k) {
    const double ang = -2.0 * M_PI * (double)k / (double)n;
    tw[k] = make_float2((float)std::cos(ang), (float)std::sin(ang));
  }
  HH_CREATE_HIP(hipMemcpy(c->d_tw, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice));
#undef HH_CREATE_HIP
  guard.c = nullptr;
  *out = c;
  return HH_OK;
} HH_CATCH_CTX(nullptr, "hh_create2")

void hh_destroy(hh_ctx* c) try {
  if (!c) return;
  (void)hh_comm_destroy(c);
  (void)hipSetDevice(c->device);
  if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
  for (auto& e : c->events) {
    (void)hipEventDestroy(e.a);
    (void)hipEventDestroy(e.b);
  }
  (void)hipFree(c->d_tw);
  (void)hipFree(c->d_inter);
  (void)hipFree(c->d_partials);
  (void)hipFree(c->d_psum);
  (void)hipFree(c->d_params);
  (void)hipFree(c->d_scores);
  (void)hipFree(c->d_units);
  (void)hipFree(c->d_w2);
  (void)hipFree(c->d_wec);
  (void)hipFree(c->d_q);
  (void)hipFree(c->d_cpart);
  (void)hipFree(c->d_ref);
  (void)hipFree(c->d_kb_list);
  (void)hipFree(c->d_seg_rows);
  (void)hipFree(c->d_q_roff);
  (void)hipFree(c->d_table);
  (void)hipFree(c->d_eg);
  (void)hipFree(c->d_cgs);
  (void)hipFree(c->d_run_imax);
  (void)hipFree(c->d_argmax);

  (void)hipFree(c->d_spec);
  (void)hipFree(c->d_img);
  gen_free(c);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
} catch (...) {
}

int hh_max_batch(const hh_ctx* c) { return c ? c->max_batch : HH_ERR_ARG; }

// Device memory the context holds right now (its buffers grow with the sweeps it has run and are kept until
// hh_destroy): the sizes the runtime reports for its allocations.  out (may be NULL): {run tables, column factors,
// two-pass intermediate, several-segment buffers (q + covariance numerators), everything else}.
int64_t hh_memory_bytes(const hh_ctx* c, int64_t out[5]) try {
  if (!c) return HH_ERR_ARG;
  (void)hipSetDevice(c->device);
  auto size_of = [](const void* p) -> int64_t {
    size_t n = 0;
    return (p && hipMemPtrGetInfo(const_cast<void*>(p), &n) == hipSuccess) ? (int64_t)n : 0;
  };
  int64_t part[5] = {0, 0, 0, 0, 0};
  part[0] = size_of(c->d_table);
  part[1] = size_of(c->d_eg) + size_of(c->d_cgs);
  part[2] = size_of(c->d_inter);
  part[3] = size_of(c->d_q) + size_of(c->d_cpart) + size_of(c->d_wec) + size_of(c->d_psum);
  for (const void* p : {(const void*)c->d_tw, (const void*)c->d_partials, (const void*)c->d_params, (const void*)c->d_scores, (const void*)c->d_units,
                        (const void*)c->d_w2, (const void*)c->d_ref, (const void*)c->d_kb_list, (const void*)c->d_run_imax, (const void*)c->d_argmax,
                        (const void*)c->d_spec, (const void*)c->d_img})
    part[4] += size_of(p);
  if (const hh_gen* g = c->gen) {
    part[0] += size_of(g->d_table);
    part[1] += size_of(g->d_eg) + size_of(g->d_cgs);
    part[3] += size_of(g->d_q) + size_of(g->d_cpart) + size_of(g->d_wec) + size_of(g->d_psum);
    for (const void* p : {(const void*)g->d_tw_nx, (const void*)g->d_tw_ny, (const void*)g->d_w2, (const void*)g->d_runs, (const void*)g->d_run_of,
                          (const void*)g->d_layers, (const void*)g->d_partials, (const void*)g->d_r, (const void*)g->d_f, (const void*)g->d_cent})
      part[4] += size_of(p);
  }
  int64_t total = 0;
  for (int k = 0; k < 5; ++k) { total += part[k]; if (out) out[k] = part[k]; }
  return total;
} HH_CATCH_CTX(c, "hh_memory_bytes")

// The fused pass's launch plan for `runs` runs of `run_len` candidates, `n_kb` ky blocks and `slots` resident workgroups
// (pure host arithmetic, no device needed): out = {runs_a, groups_a, cpw_a, groups_b, cpw_b, layers}.
int hh_general_plan(int nx, int rows_lds, int kg, int64_t out[6]) try {
  if (!out || nx < 1 || rows_lds < 1 || kg < 1 || kg > 32) return HH_ERR_ARG;
  GenRowsPlan rp{};
  const bool ok = gen_rows_plan(nx, rows_lds, kg, &rp);
  out[0] = ok ? rp.r1 : 0;
  out[1] = ok ? rp.r2 : 0;
  out[2] = ok ? rp.rows_per_block : 8;
  out[3] = ok ? rp.threads : GEN_THREADS;
  out[4] = ok ? rp.halves : 1;
  out[5] = ok ? (int64_t)rp.lds : 0;
  return HH_OK;
} HH_CATCH_CTX(nullptr, "hh_general_plan")

int hh_fused_schedule(int64_t runs, int run_len, int n_kb, int slots, int32_t out[6]) try {
  if (!out || runs < 1 || run_len < 1 || n_kb < 1 || slots < 1) return HH_ERR_ARG;
  const FusedSchedule fs = fused_schedule(runs, run_len, n_kb, slots);
  out[0] = fs.runs_a; out[1] = fs.groups_a; out[2] = fs.cpw_a; out[3] = fs.groups_b; out[4] = fs.cpw_b; out[5] = fs.layers;
  return HH_OK;
} HH_CATCH_CTX(nullptr, "hh_fused_schedule")

int hh_set_stream(hh_ctx* c, void* hip_stream) try {
  if (!c) return HH_ERR_ARG;
  HH_HIP(c, hipSetDevice(c->device));
  HH_HIP(c, hipStreamSynchronize(c->stream));
  c->stream = reinterpret_cast<hipStream_t>(hip_stream);  // NULL is the device's null stream, a valid choice
  return HH_OK;
} HH_CATCH_CTX(c, "hh_set_stream")

int hh_use_own_stream(hh_ctx* c) try {
  if (!c) return HH_ERR_ARG;
  HH_HIP(c, hipSetDevice(c->device));
  HH_HIP(c, hipStreamSynchronize(c->stream));
  c->stream = c->own_stream;
  return HH_OK;
} HH_CATCH_CTX(nullptr, "hh_use_own_stream")

int hh_synchronize(hh_ctx* c) try {
  if (!c) return HH_ERR_ARG;
  HH_HIP(c, hipSetDevice(c->device));
  HH_HIP(c, hipStreamSynchronize(c->stream));
  return HH_OK;
} HH_CATCH_CTX(nullptr, "hh_synchronize")

int hh_set_geometry(hh_ctx* c, const hh_geom* g) try {
  if (!c || !g) return fail(c, HH_ERR_ARG, "hh_set_geometry: NULL argument");
  if (!(g->apix > 0) || !(g->ball_radius > 0) || !(g->helical_diameter >= 0))
    return fail(c, HH_ERR_ARG, "hh_set_geometry: apix and ball_radius must be positive");
  // utils.py:88
  if (!(g->helical_diameter + g->ball_radius < c->ny * g->apix * 0.99))
    return fail(c, HH_ERR_ARG, "hh_set_geometry: helical_diameter + ball_radius must be < 0.99 * ny * apix");
  if (g->n_units < 0 || g->n_units > HH_MAX_UNITS) return fail(c, HH_ERR_ARG, "hh_set_geometry: n_units out of range");
  if (g->n_units > 1 && !g->units) return fail(c, HH_ERR_ARG, "hh_set_geometry: units is NULL");
  HH_HIP(c, hipSetDevice(c->device));
  DevGeom d{};
  d.height = (double)c->nx * g->apix;
  d.apix = (float)g->apix;
  d.inv_apix = (float)(1.0 / g->apix);
  const double sigma2 = g->ball_radius * g->ball_radius / std::log(2.0);
  d.inv_sigma2 = (float)(1.0 / sigma2);
  d.dy = g->dy;
  const int bits = g->tail_bits > 0 ? g->tail_bits : 24;
  // exp(-(R*apix)^2 / sigma2) < 2^-bits
  d.rpx = (int)std::ceil(std::sqrt(sigma2 * bits * std::log(2.0)) / g->apix);
  if (d.rpx < 1) d.rpx = 1;
  if (d.rpx > std::max(c->ny, c->nx)) d.rpx = std::max(c->ny, c->nx);
  d.has_rot = (g->tilt != 0.0 || g->psi != 0.0) ? 1 : 0;
  {
    // R = Rx(-psi) * Ry(tilt) (scipy from_euler("yx", (tilt, -psi)), utils.py:167); rows 1, 2
    const double a = g->tilt * M_PI / 180.0, b = -g->psi * M_PI / 180.0;
    const double ca = std::cos(a), sa = std::sin(a), cb = std::cos(b), sb = std::sin(b);
    d.m[0] = sb * sa;  d.m[1] = cb; d.m[2] = -sb * ca;
    d.m[3] = -cb * sa; d.m[4] = sb; d.m[5] = cb * ca;
  }
  std::vector<double> units;
  if (g->n_units <= 1 && !g->units) {
    units = {g->helical_diameter / 2.0, 0.0, 0.0};
    d.n_units = 1;
  } else {
    d.n_units = std::max(1, g->n_units);
    units.assign(g->units, g->units + 3 * (size_t)d.n_units);
  }
  {
    double rmax = 0, zabs = 0;
    for (int u = 0; u < d.n_units; ++u) {
      rmax = std::max(rmax, std::fabs(units[3 * u]));
      zabs = std::max(zabs, std::fabs(units[3 * u + 2]));
    }
    const double m3 = d.has_rot ? d.m[3] : 0.0, m4 = d.has_rot ? d.m[4] : 0.0, m5 = d.has_rot ? d.m[5] : 1.0;
    if (!d.has_rot) { d.m[0] = 0; d.m[1] = 1; d.m[2] = 0; d.m[3] = 0; d.m[4] = 0; d.m[5] = 1; }
    d.fast = std::fabs(m5) >= 0.25 ? 1 : 0;
    d.slack = (float)(rmax * (std::fabs(m3) + std::fabs(m4)) + zabs * std::fabs(m5) + 1e-3);
  }
  HH_HIP(c, hipMemcpyAsync(c->d_units, units.data(), units.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HH_HIP(c, hipStreamSynchronize(c->stream));
  c->geom = d;
  c->have_geom = true;
  return HH_OK;
} HH_CATCH_CTX(c, "hh_set_geometry")

int hh_set_reference(hh_ctx* c, const float* images, int n_segments, const uint8_t* mask, int log_flag) try {
  if (!c || !images || !mask || n_segments <= 0) return fail(c, HH_ERR_ARG, "hh_set_reference: bad argument");
  HH_HIP(c, hipSetDevice(c->device));
  if (c->general) return gen_set_reference(c, images, n_segments, mask, log_flag);
  const int n = c->n;
  const size_t npix = (size_t)n * n, nh = (size_t)(n / 2 + 1) * n;
  int rc = ensure_img(c, n_segments);
  if (rc) return rc;
  HH_HIP(c, hipMemcpyAsync(c->d_img, images, (size_t)n_segments * npix * sizeof(float), hipMemcpyHostToDevice, c->stream));
  rc = spectra_of_images(c, n_segments);
  if (rc) return rc;
  std::vector<float2> spec((size_t)n_segments * nh);
  HH_HIP(c, hipMemcpyAsync(spec.data(), c->d_spec, spec.size() * sizeof(float2), hipMemcpyDeviceToHost, c->stream));
  HH_HIP(c, hipStreamSynchronize(c->stream));

  // Hermitian half-plane weights: W[ky][kx] = mask(k) + mask(-k) for 0 < ky < N/2, mask(k) on
  // the two self-paired rows ky = 0 and ky = N/2 (both members of a pair lie in the half plane).
  std::vector<float> w(nh);
  double sw = 0;
  auto mu = [&](int uy, int ux) {  // mask at unshifted frequency indices
    return mask[(size_t)((uy + n / 2) & (n - 1)) * n + ((ux + n / 2) & (n - 1))] ? 1.f : 0.f;
  };
  for (int ky = 0; ky <= n / 2; ++ky)
    for (int kx = 0; kx < n; ++kx) {
      float v = mu(ky, kx);
      if (ky > 0 && ky < n / 2) v += mu((n - ky) & (n - 1), (n - kx) & (n - 1));
      w[(size_t)ky * n + kx] = v;
      sw += v;
    }
  if (!(sw > 0)) return fail(c, HH_ERR_ARG, "hh_set_reference: the mask selects no Fourier bin");
  // ky blocks (8 rows of the intermediate) that carry any weight; block 0 also holds the packed
  // ky = 0 / ky = N/2 row and is always kept.  K_A does not store the others, K_B does not read them.
  std::vector<int> kb_list;
  unsigned long long kb_mask = 0;
  for (int kb = 0; kb < n / 16; ++kb) {
    bool any = kb == 0;
    for (int r = 8 * kb; r < 8 * kb + 8 && !any; ++r)
      for (int kx = 0; kx < n && !any; ++kx) any = w[(size_t)r * n + kx] > 0.f;
    if (any) {
      kb_list.push_back(kb);
      kb_mask |= 1ull << kb;
    }
  }

  // device tables: W2 (weights, and for one segment the centred spectrum, in K_B's lane-major order);
  // with several segments the centred spectra go to a [Sp][K] matrix for the MFMA contraction
  const bool multi = n_segments > 1;
  const int s_pad = multi ? (n_segments + 63) / 64 * 64 : 0;
  // several segments: the shared-twist pipelines batch up to HH_SEG_BATCH candidates (q of a batch lives in HBM:
  // 0.5 MB per candidate at N = 512), the general pipeline max_batch
  std::vector<float2> w2(nh);
  // several segments, N <= 512: q and the centred spectra keep only the bins with weight (a fifth fewer bytes through HBM
  // under the radial band, most of them under a resolution-limited or layer-line mask): row r's bins from roff[r] on, in
  // ascending kx — the order the kernels' ballots give (compact_positions)
  const bool compact = multi && n <= 512 && std::getenv("HH_Q_FULL") == nullptr;
  std::vector<int> roff((size_t)n / 2 + 2, 0), cpos(compact ? nh : 0, -1);
  size_t q_stride = nh;
  if (compact) {
    int at = 0;
    for (int r = 0; r <= n / 2; ++r) {
      roff[(size_t)r] = at;
      for (int kx = 0; kx < n; ++kx)
        if (w[(size_t)r * n + kx] > 0.f) cpos[(size_t)r * n + kx] = at++;
    }
    roff[(size_t)n / 2 + 1] = at;
    q_stride = ((size_t)at + 511) / 512 * 512;
    if (q_stride == 0) q_stride = 512;
  }
  std::vector<float> wecm(multi ? (size_t)s_pad * q_stride : 0, 0.f);
  c->ref.assign(n_segments, RefConsts{});
  std::vector<double> e(nh);
  for (int s = 0; s < n_segments; ++s) {
    const float2* sp = spec.data() + (size_t)s * nh;
    double se = 0;
    for (size_t i = 0; i < nh; ++i) {
      const double a = std::hypot((double)sp[i].x, (double)sp[i].y);
      e[i] = log_flag ? std::log1p(a) : a;
      se += (double)w[i] * e[i];
    }
    const double ebar = se / sw;
    double swec = 0, var_e = 0;
    for (size_t i = 0; i < nh; ++i) {
      const double dc = e[i] - ebar;
      const float wec = (float)((double)w[i] * dc);
      if (s == 0) {  // device layout: within a row, lane t of the row's FFT finds its bins kx = t + T m at [t][m]
        const size_t row = i / n, kx = i % n, tt = kx % (n / 8), mm = kx / (n / 8);
        w2[row * n + tt * 8 + mm] = make_float2(w[i], wec);
      }
      if (multi && compact) {
        if (cpos[i] >= 0) wecm[(size_t)s * q_stride + (size_t)cpos[i]] = wec;
      } else if (multi) {   // in q's order: bin kx = wl + m (n / 8) of a row at [wl][m]
        const size_t row = i / n, kx = i % n;
        wecm[(size_t)s * nh + row * n + (kx % (n / 8)) * 8 + kx / (n / 8)] = wec;
      }
      swec += (double)wec;
      var_e += (double)w[i] * dc * dc;
    }
    c->ref[s] = RefConsts{sw, swec, var_e};
  }
  // From here on the old reference is gone: a failure below leaves the context in the "no reference" state (the next
  // sweep returns HH_ERR_STATE) instead of a mix of old sizes and new tables.  The device tables are kept between
  // calls and only grow.
  c->n_segments = 0;
  int rcg = ensure_bytes(c, (void**)&c->d_w2, &c->cap_w2, w2.size() * sizeof(float2));
  if (rcg) return rcg;
  HH_HIP(c, hipMemcpyAsync(c->d_w2, w2.data(), w2.size() * sizeof(float2), hipMemcpyHostToDevice, c->stream));
  if (multi) {
    std::vector<int> seg_rows;
    for (int r = 0; r <= n / 2; ++r) {
      bool any = false;
      for (int kx = 0; kx < n && !any; ++kx) any = w[(size_t)r * n + kx] > 0.f;
      if (any) seg_rows.push_back(r);
    }
    if ((rcg = ensure_bytes(c, (void**)&c->d_seg_rows, &c->cap_seg_rows, (size_t)(n / 2 + 1) * sizeof(int)))) return rcg;
    HH_HIP(c, hipMemcpyAsync(c->d_seg_rows, seg_rows.data(), seg_rows.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    c->n_seg_rows = (int)seg_rows.size();
    if (compact) {
      if ((rcg = ensure_bytes(c, (void**)&c->d_q_roff, &c->cap_q_roff, roff.size() * sizeof(int)))) return rcg;
      HH_HIP(c, hipMemcpyAsync(c->d_q_roff, roff.data(), roff.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
      HH_HIP(c, hipStreamSynchronize(c->stream));   // (roff is a local)
    }
    if (q_stride != c->q_stride) c->b_pad = 0;   // the buffers of a batch are laid out by it: start over
    c->q_stride = q_stride;
    c->q_slices = compact ? (int)(q_stride / 512) : 0;
    if ((rcg = ensure_bytes(c, (void**)&c->d_wec, &c->cap_wec, wecm.size() * sizeof(float)))) return rcg;
    if ((rcg = ensure_bytes(c, (void**)&c->d_ref, &c->cap_ref, (size_t)n_segments * sizeof(RefConsts)))) return rcg;
    HH_HIP(c, hipMemcpyAsync(c->d_wec, wecm.data(), wecm.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HH_HIP(c, hipMemcpyAsync(c->d_ref, c->ref.data(), (size_t)n_segments * sizeof(RefConsts), hipMemcpyHostToDevice,
                             c->stream));
  }
  if ((rcg = ensure_bytes(c, (void**)&c->d_kb_list, &c->cap_kb, (size_t)(n / 16) * sizeof(int)))) return rcg;
  HH_HIP(c, hipMemcpyAsync(c->d_kb_list, kb_list.data(), kb_list.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
  // rows of skipped ky blocks keep zero moments
  HH_HIP(c, hipMemsetAsync(c->d_partials, 0, (size_t)2 * c->cap_partials * npart_for(n) * 3 * sizeof(double), c->stream));
  c->n_kb = (int)kb_list.size();
  c->kb_mask = kb_mask;
  if (s_pad != c->s_pad) {   // the covariance buffer's row length changes with the segment count: start over
    c->b_pad = 0;
  }
  c->s_pad = s_pad;
  if (multi && (rcg = ensure_segment_buffers(c, c->max_batch))) return rcg;   // the general pipeline's batches; shared-twist sweeps grow it
  HH_HIP(c, hipStreamSynchronize(c->stream));
  c->n_segments = n_segments;
  c->log_flag = log_flag ? 1 : 0;
  return HH_OK;
} HH_CATCH_CTX(c, "hh_set_reference")

int hh_sweep_device(hh_ctx* c, const double* d_params, int64_t g, float* d_scores) try {
  int rc = check_ready(c, true);
  if (rc) return rc;
  if (!d_params || !d_scores || g < 0) return fail(c, HH_ERR_ARG, "hh_sweep_device: bad argument");
  if (g == 0) return HH_OK;
  HH_HIP(c, hipSetDevice(c->device));
  return sweep_on_device(c, d_params, g, d_scores);
} HH_CATCH_CTX(c, "hh_sweep_device")

int hh_sweep_device_mirrored(hh_ctx* c, const double* d_params, const double* h_params, int64_t g, float* d_scores) try {
  int rc = check_ready(c, true);
  if (rc) return rc;
  if (!d_params || !d_scores || g < 0) return fail(c, HH_ERR_ARG, "hh_sweep_device_mirrored: bad argument");
  if (g == 0) return HH_OK;
  HH_HIP(c, hipSetDevice(c->device));
  return sweep_on_device(c, d_params, g, d_scores, h_params);
} HH_CATCH_CTX(c, "hh_sweep_device_mirrored")

int hh_sweep_device_strided(hh_ctx* c, const double* d_params, const double* h_params, int64_t g, float* d_scores,
                            int64_t ld_scores) try {
  int rc = check_ready(c, true);
  if (rc) return rc;
  if (!d_params || !d_scores || g < 0 || ld_scores < g) return fail(c, HH_ERR_ARG, "hh_sweep_device_strided: bad argument");
  if (g == 0) return HH_OK;
  HH_HIP(c, hipSetDevice(c->device));
  return sweep_on_device(c, d_params, g, d_scores, h_params, ld_scores);
} HH_CATCH_CTX(c, "hh_sweep_device_strided")

int hh_last_first_pass(const hh_ctx* c) { return c ? c->last_first_pass : HH_ERR_ARG; }

int hh_set_table_path(hh_ctx* c, int mode) try {
  if (!c) return HH_ERR_ARG;
  c->table_path = mode ? 1 : 0;
  c->fused_path = mode == 1 ? 0 : 1;  // 1: run tables through the two-pass pipeline; 2 (default): fused where it fits
  return HH_OK;
} HH_CATCH_CTX(c, "hh_set_table_path")

int hh_sweep(hh_ctx* c, const double* params, int64_t g, float* scores) try {
  int rc = check_ready(c, true);
  if (rc) return rc;
  if (!params || !scores || g < 0) return fail(c, HH_ERR_ARG, "hh_sweep: bad argument");
  if (g == 0) return HH_OK;
  HH_HIP(c, hipSetDevice(c->device));
  if (g > c->cap_params) {
    if (c->d_params) HH_HIP(c, hipFree(c->d_params));
    c->d_params = nullptr;
    c->cap_params = 0;
    HH_HIP(c, hipMalloc(&c->d_params, (size_t)g * 4 * sizeof(double)));
    c->cap_params = g;
  }
  // staging for the scores of all segments (S x G); kept between calls
  const int64_t need_scores = g * c->n_segments;
  if (need_scores > c->cap_scores) {
    if (c->d_scores) HH_HIP(c, hipFree(c->d_scores));
    c->d_scores = nullptr;
    c->cap_scores = 0;
    HH_HIP(c, hipMalloc(&c->d_scores, (size_t)need_scores * sizeof(float)));
    c->cap_scores = need_scores;
  }
  float* const d_sc = c->d_scores;
  hipError_t e = hipMemcpyAsync(c->d_params, params, (size_t)g * 4 * sizeof(double), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    rc = sweep_on_device(c, c->d_params, g, d_sc, params);
    if (rc == HH_OK) {
      e = hipMemcpyAsync(scores, d_sc, (size_t)g * c->n_segments * sizeof(float), hipMemcpyDeviceToHost, c->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
  }
  if (rc) return rc;
  if (e != hipSuccess) return fail(c, HH_ERR_HIP, std::string("hh_sweep: ") + hipGetErrorString(e));
  return HH_OK;
} HH_CATCH_CTX(c, "hh_sweep")

int hh_argmax(const float* scores, int64_t n, int64_t* index) try {
  if (!scores || !index || n <= 0) return HH_ERR_ARG;
  int64_t best = -1;
  for (int64_t i = 0; i < n; ++i) {
    if (scores[i] != scores[i]) continue;  // NaN never wins
    if (best < 0 || scores[i] > scores[best]) best = i;
  }
  *index = best < 0 ? 0 : best;
  return HH_OK;
} HH_CATCH_CTX(nullptr, "hh_argmax")

int hh_argmax_device(hh_ctx* c, const float* d_scores, int64_t n_rows, int64_t n, int64_t ld, int64_t* d_index,
                     int64_t* h_index) try {
  if (!c || !d_scores || (!d_index && !h_index) || n_rows <= 0 || n <= 0 || n_rows > 65535 || (ld != 0 && ld < n))
    return fail(c, HH_ERR_ARG, "hh_argmax_device: bad argument");
  HH_HIP(c, hipSetDevice(c->device));
  if (!d_index) {
    int rc = ensure_bytes(c, (void**)&c->d_argmax, &c->cap_argmax, (size_t)n_rows * sizeof(int64_t));
    if (rc) return rc;
    d_index = c->d_argmax;
  }
  hipLaunchKernelGGL(k_argmax_rows, dim3((unsigned)n_rows), dim3(1024), 0, c->stream, d_scores, n, ld ? ld : n, d_index);
  HH_HIP(c, hipGetLastError());
  if (h_index) {
    HH_HIP(c, hipMemcpyAsync(h_index, d_index, (size_t)n_rows * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    HH_HIP(c, hipStreamSynchronize(c->stream));
  }
  return HH_OK;
} HH_CATCH_CTX(c, "hh_argmax_device")

// ---- the one collective of a multi-GPU sweep, without a host framework: RCCL through dlopen --------------------
namespace {
struct Rccl {
  struct UniqueId { char b[128]; };  // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128), passed by value
  void* lib = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
Rccl* rccl(std::string& err) {
  static Rccl r;
  static std::once_flag once;   // two threads may make their first hh_comm_* call together
  std::call_once(once, [] {
    // the soname first: a process that already holds RCCL (torch.distributed's "nccl" backend) gets that same copy
    for (const char* name : {"librccl.so.1", "librccl.so"})
      if ((r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
    if (r.lib) {
      r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.lib, "ncclGetUniqueId"));
      r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.lib, "ncclCommInitRank"));
      r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.lib, "ncclAllGather"));
      r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
      r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
    }
  });
  if (!r.lib || !r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy) {
    err = "librccl.so.1 is not loadable (the multi-GPU entry points need RCCL)";
    return nullptr;
  }
  return &r;
}
}  // namespace

int hh_comm_unique_id(void* id128) try {
  if (!id128) return fail(nullptr, HH_ERR_ARG, "hh_comm_unique_id: NULL");
  std::string err;
  Rccl* r = rccl(err);
  if (!r) return fail(nullptr, HH_ERR_STATE, err);
  const int rc = r->GetUniqueId(id128);
  if (rc) return fail(nullptr, HH_ERR_HIP, std::string("ncclGetUniqueId: ") + (r->GetErrorString ? r->GetErrorString(rc) : "?"));
  return HH_OK;
} HH_CATCH_CTX(nullptr, "hh_comm_unique_id")

int hh_comm_init(hh_ctx* c, int rank, int world, const void* id128) try {
  if (!c || !id128 || world < 1 || rank < 0 || rank >= world) return fail(c, HH_ERR_ARG, "hh_comm_init: bad argument");
  if (c->comm) return fail(c, HH_ERR_STATE, "hh_comm_init: the context already has a communicator");
  std::string err;
  Rccl* r = rccl(err);
  if (!r) return fail(c, HH_ERR_STATE, err);
  HH_HIP(c, hipSetDevice(c->device));
  Rccl::UniqueId id;
  std::memcpy(id.b, id128, sizeof(id.b));
  const int rc = r->CommInitRank(&c->comm, world, id, rank);
  if (rc) {
    c->comm = nullptr;
    return fail(c, HH_ERR_HIP, std::string("ncclCommInitRank: ") + (r->GetErrorString ? r->GetErrorString(rc) : "?"));
  }
  return HH_OK;
} HH_CATCH_CTX(c, "hh_comm_init")

int hh_allgather(hh_ctx* c, const float* d_send, int64_t count, float* d_recv) try {
  if (!c || !d_send || !d_recv || count <= 0) return fail(c, HH_ERR_ARG, "hh_allgather: bad argument");
  if (!c->comm) return fail(c, HH_ERR_STATE, "hh_allgather: hh_comm_init has not been called");
  std::string err;
  Rccl* r = rccl(err);
  if (!r) return fail(c, HH_ERR_STATE, err);
  HH_HIP(c, hipSetDevice(c->device));
  const int rc = r->AllGather(d_send, d_recv, (size_t)count, /* ncclFloat32 */ 7, c->comm, c->stream);
  if (rc) return fail(c, HH_ERR_HIP, std::string("ncclAllGather: ") + (r->GetErrorString ? r->GetErrorString(rc) : "?"));
  return HH_OK;
} HH_CATCH_CTX(c, "hh_allgather")

int hh_comm_destroy(hh_ctx* c) try {
  if (!c) return HH_ERR_ARG;
  if (!c->comm) return HH_OK;
  std::string err;
  Rccl* r = rccl(err);
  if (r) {
    (void)hipSetDevice(c->device);
    // a borrowed stream (hh_set_stream) may be gone by the time the context is torn down: only the library's own
    // stream is waited for; a caller that queued a collective on ITS stream synchronises that stream itself
    if (c->stream == c->own_stream && c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    (void)r->CommDestroy(c->comm);
  }
  c->comm = nullptr;
  return HH_OK;
} HH_CATCH_CTX(nullptr, "hh_comm_destroy")

int hh_simulate(hh_ctx* c, const double* params, float* image_out) try {
  int rc = check_ready(c, false);
  if (rc) return rc;
  if (!params || !image_out) return fail(c, HH_ERR_ARG, "hh_simulate: bad argument");
  HH_HIP(c, hipSetDevice(c->device));
  if (c->general) return gen_simulate(c, params, image_out);
  const size_t npix = (size_t)c->n * c->n;
  rc = ensure_img(c, 1);
  if (rc) return rc;
  double* dp = nullptr;
  HH_HIP(c, hipMalloc(&dp, 4 * sizeof(double)));
  hipError_t e = hipMemcpyAsync(dp, params, 4 * sizeof(double), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    FirstArgs fa{};
    fa.params = dp;
    fa.units = c->d_units;
    fa.twtab = c->d_tw;
    fa.inter = c->d_inter;
    fa.raster_out = c->d_img;
    fa.kb_mask = ~0ull;
    fa.g = c->geom;
    rc = dispatch_first<MODE_RASTER_OUT>(c, fa, 1);
    if (rc == HH_OK) {
      e = hipMemcpyAsync(image_out, c->d_img, npix * sizeof(float), hipMemcpyDeviceToHost, c->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
  }
  (void)hipFree(dp);
  if (rc) return rc;
  if (e != hipSuccess) return fail(c, HH_ERR_HIP, std::string("hh_simulate: ") + hipGetErrorString(e));
  return HH_OK;
} HH_CATCH_CTX(c, "hh_simulate")

int hh_power_spectrum(hh_ctx* c, const float* image, int log_flag, float* pwr_out, float* phase_out) try {
  if (!c || !image || !pwr_out) return fail(c, HH_ERR_ARG, "hh_power_spectrum: bad argument");
  HH_HIP(c, hipSetDevice(c->device));
  if (c->general) return gen_power_spectrum(c, image, log_flag, pwr_out, phase_out);
  const int n = c->n;
  const size_t npix = (size_t)n * n;
  int rc = ensure_img(c, 3);  // [0] input, [1] pwr, [2] phase
  if (rc) return rc;
  HH_HIP(c, hipMemcpyAsync(c->d_img, image, npix * sizeof(float), hipMemcpyHostToDevice, c->stream));
  rc = spectra_of_images(c, 1);
  if (rc) return rc;
  unsigned* d_mm = nullptr;
  HH_HIP(c, hipMalloc(&d_mm, 2 * sizeof(unsigned)));
  const unsigned init[2] = {0x7f800000u, 0u};  // +inf, 0
  hipError_t e = hipMemcpyAsync(d_mm, init, sizeof(init), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    float* d_pwr = c->d_img + npix;
    float* d_phase = phase_out ? c->d_img + 2 * npix : nullptr;
    hipLaunchKernelGGL(k_expand_spectrum, dim3(n), dim3(256), 0, c->stream, c->d_spec, n, log_flag ? 1 : 0, d_pwr,
                       d_phase, d_mm);
    hipLaunchKernelGGL(k_normalise, dim3(256), dim3(256), 0, c->stream, d_pwr, npix, d_mm);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(pwr_out, d_pwr, npix * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && phase_out)
      e = hipMemcpyAsync(phase_out, d_phase, npix * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  }
  (void)hipFree(d_mm);
  if (e != hipSuccess) return fail(c, HH_ERR_HIP, std::string("hh_power_spectrum: ") + hipGetErrorString(e));
  return HH_OK;
} HH_CATCH_CTX(c, "hh_power_spectrum")

int hh_low_high_pass_filter(hh_ctx* c, const float* image, double low_pass_fraction, double high_pass_fraction,
                            float* out) try {
  if (!c || !image || !out) return fail(c, HH_ERR_ARG, "hh_low_high_pass_filter: bad argument");
  HH_HIP(c, hipSetDevice(c->device));
  if (c->general) return gen_low_high_pass_filter(c, image, low_pass_fraction, high_pass_fraction, out);
  const int n = c->n;
  const size_t npix = (size_t)n * n, nh = (size_t)(n / 2 + 1) * n;
  int rc = ensure_img(c, 2);  // [0] input, [1] output
  if (rc) return rc;
  rc = ensure_spec(c, 2);     // [0] spectrum, [1] rows after the inverse transform along x
  if (rc) return rc;
  HH_HIP(c, hipMemcpyAsync(c->d_img, image, npix * sizeof(float), hipMemcpyHostToDevice, c->stream));
  rc = spectra_of_images(c, 1);
  if (rc) return rc;
  // filters.py:362-369: each filter applies only for a fraction strictly inside (0, 1)
  const bool lp = low_pass_fraction > 0 && low_pass_fraction < 1, hp = high_pass_fraction > 0 && high_pass_fraction < 1;
  const float f2_lp = lp ? (float)(std::log(2.0) / (low_pass_fraction * low_pass_fraction)) : 0.f;
  const float f2_hp = hp ? (float)(std::log(2.0) / (high_pass_fraction * high_pass_fraction)) : 0.f;
  float* const d_out = c->d_img + npix;
#define HH_CALL(NN) launch_filter<NN>(c, f2_lp, f2_hp, c->d_spec, c->d_spec + nh, d_out)
  rc = [&]() -> int { HH_SWITCH_N(c, HH_CALL); }();
#undef HH_CALL
  if (rc) return rc;
  HH_HIP(c, hipMemcpyAsync(out, d_out, npix * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HH_HIP(c, hipStreamSynchronize(c->stream));
  return HH_OK;
} HH_CATCH_CTX(c, "hh_low_high_pass_filter")

int hh_threshold_data(hh_ctx* c, const float* data, int64_t n, int use_fraction, double thresh, float* out) try {
  if (!c || !data || !out || n <= 0) return fail(c, HH_ERR_ARG, "hh_threshold_data: bad argument");
  HH_HIP(c, hipSetDevice(c->device));
  float *d_x = nullptr, *d_y = nullptr;
  int* d_mx = nullptr;
  HH_HIP(c, hipMalloc(&d_x, (size_t)n * sizeof(float)));
  hipError_t e = hipMalloc(&d_y, (size_t)n * sizeof(float));
  if (e == hipSuccess) e = hipMalloc(&d_mx, 2 * sizeof(int));
  const int init[2] = {(int)0x80000000, -1};  // below every non-negative float's bits; above every negative float's (as unsigned)
  if (e == hipSuccess) e = hipMemcpyAsync(d_x, data, (size_t)n * sizeof(float), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_mx, init, sizeof(init), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    const int grid = (int)std::min<int64_t>(1024, (n + 255) / 256);
    if (use_fraction) hipLaunchKernelGGL(k_array_max, dim3(grid), dim3(256), 0, c->stream, d_x, (size_t)n, d_mx);
    hipLaunchKernelGGL(k_threshold, dim3(grid), dim3(256), 0, c->stream, d_x, (size_t)n, d_mx, (float)thresh, (float)thresh,
                       use_fraction ? 1 : 0, d_y);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(out, d_y, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(d_x);
  (void)hipFree(d_y);
  (void)hipFree(d_mx);
  if (e != hipSuccess) return fail(c, HH_ERR_HIP, std::string("hh_threshold_data: ") + hipGetErrorString(e));
  return HH_OK;
} HH_CATCH_CTX(c, "hh_threshold_data")

int hh_cross_correlation(hh_ctx* c, const float* a, const float* b, int64_t n, double* out) try { return pearson(c, a, b, n, out); } HH_CATCH_CTX(c, "hh_cross_correlation")
int hh_cross_correlation_f64(hh_ctx* c, const double* a, const double* b, int64_t n, double* out) try { return pearson(c, a, b, n, out); } HH_CATCH_CTX(c, "hh_cross_correlation_f64")
int hh_cosine_similarity(hh_ctx* c, const float* a, const float* b, int64_t n, double* out) try { return cosine(c, a, b, n, out); } HH_CATCH_CTX(c, "hh_cosine_similarity")
int hh_cosine_similarity_f64(hh_ctx* c, const double* a, const double* b, int64_t n, double* out) try { return cosine(c, a, b, n, out); } HH_CATCH_CTX(c, "hh_cosine_similarity_f64")

}  // extern "C"

namespace {
// scipy.ndimage.affine_transform(data, matrix, offset, order = 1, mode = "constant", cval = 0) of a 2-D image — what
// helicon.rotate_shift_image (lib/transforms.py:315-369) calls: output pixel (y, x) samples the input at
// c = M (y, x) + offset in float64; a coordinate outside [0, n - 1] on either axis gives 0 (ni_interpolation.c maps it
// to -1 and takes the constant); inside, the two-point linear weights on floor(c), floor(c) + 1 (an index past the
// edge only ever carries weight 0).
__global__ __launch_bounds__(256) void k_affine_bilinear(const float* __restrict__ in, int ny, int nx, double m00, double m01,
                                                         double m10, double m11, double o0, double o1, float* __restrict__ out) {
#pragma clang fp contract(off)
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= nx) return;
  const double cy = m00 * (double)y + m01 * (double)x + o0;
  const double cx = m10 * (double)y + m11 * (double)x + o1;
  float v = 0.f;
  if (cy >= 0.0 && cy <= (double)(ny - 1) && cx >= 0.0 && cx <= (double)(nx - 1)) {
    const double fy = floor(cy), fx = floor(cx);
    const int y0 = (int)fy, x0 = (int)fx;
    const double wy = cy - fy, wx = cx - fx;
    const int y1 = min(y0 + 1, ny - 1), x1 = min(x0 + 1, nx - 1);
    // scipy sums over the 2 x 2 support in (y, x) order with weights w_y w_x
    double acc = 0.0;
    acc += (double)in[(size_t)y0 * nx + x0] * ((1.0 - wy) * (1.0 - wx));
    acc += (double)in[(size_t)y0 * nx + x1] * ((1.0 - wy) * wx);
    acc += (double)in[(size_t)y1 * nx + x0] * (wy * (1.0 - wx));
    acc += (double)in[(size_t)y1 * nx + x1] * (wy * wx);
    v = (float)acc;
  }
  out[(size_t)y * nx + x] = v;
}

// ---- helicon.transform_map (lib/transforms.py:168-235): scipy.ndimage.map_coordinates(order = 3) of a volume ----------
// Step 1, the B-spline prefilter (scipy.ndimage.spline_filter, ni_splines.c): along every axis in turn, the cubic spline's
// one pole z = sqrt(3) - 2, gain (1 - z)(1 - 1/z), a causal and an anti-causal recursion with MIRROR initialisation
// ("constant" is filtered like "mirror").  One thread per line; float64 like SciPy.
__global__ __launch_bounds__(128) void k_spline_prefilter(double* __restrict__ c, int64_t n_lines, int len, int64_t inner,
                                                          int64_t stride) {
#pragma clang fp contract(off)
  // line l of an axis with `inner` elements after it: first element at (l / inner) * len * inner + l % inner, step `stride`
  const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= n_lines || len < 2) return;
  double* const p = c + (l / inner) * (int64_t)len * inner + l % inner;
  const double z = sqrt(3.0) - 2.0;
  const double gain = (1.0 - z) * (1.0 - 1.0 / z);
  for (int i = 0; i < len; ++i) p[i * stride] *= gain;
  const double zn1 = pow(z, (double)(len - 1));
  double c0 = p[0] + zn1 * p[(int64_t)(len - 1) * stride];
  double zi = z;
  for (int i = 1; i < len - 1; ++i) {
    c0 = c0 + zi * (p[i * stride] + zn1 * p[(int64_t)(len - 1 - i) * stride]);
    zi *= z;
  }
  p[0] = c0 / (1.0 - zn1 * zn1);
  for (int i = 1; i < len; ++i) p[i * stride] += z * p[(i - 1) * stride];
  p[(int64_t)(len - 1) * stride] = (z * p[(int64_t)(len - 2) * stride] + p[(int64_t)(len - 1) * stride]) * z / (z * z - 1.0);
  for (int i = len - 2; i >= 0; --i) p[i * stride] = z * (p[(i + 1) * stride] - p[i * stride]);
}

struct MapArgs {
  int nz, ny, nx;
  double m[9];            // rotation (row-major): p = m (X, Y, Z)
  double scale;
  double ox, oy, oz;      // nx // 2 - dx, ny // 2 - dy, nz // 2 - dz
};

__device__ __forceinline__ int spline_mirror(int idx, int n) {  // tap index outside [0, n - 1]: mirrored about the end points
  if (n <= 1) return 0;
  const int s2 = 2 * n - 2;
  idx = abs(idx) % s2;
  return idx >= n ? s2 - idx : idx;
}

// Step 2: output voxel (k, j, i) samples the coefficient volume at m ((i - nx/2), (j - ny/2), (k - nz/2)) scale + offset:
// 0 if the point leaves [0, n - 1] on any axis, else the 4 x 4 x 4 cubic B-spline taps (weights as ni_interpolation.c
// forms them: the last one by subtraction), summed z-outer, x-inner.
__global__ __launch_bounds__(256) void k_map_cubic(const double* __restrict__ c, MapArgs a, float* __restrict__ out) {
#pragma clang fp contract(off)
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)a.nz * a.ny * a.nx;
  if (t >= total) return;
  const int i = (int)(t % a.nx), j = (int)((t / a.nx) % a.ny), k = (int)(t / ((int64_t)a.nx * a.ny));
  double X = (double)(i - a.nx / 2), Y = (double)(j - a.ny / 2), Z = (double)(k - a.nz / 2);
  if (a.scale != 1.0) { X *= a.scale; Y *= a.scale; Z *= a.scale; }
  const double px = a.m[0] * X + a.m[1] * Y + a.m[2] * Z + a.ox;
  const double py = a.m[3] * X + a.m[4] * Y + a.m[5] * Z + a.oy;
  const double pz = a.m[6] * X + a.m[7] * Y + a.m[8] * Z + a.oz;
  float v = 0.f;
  if (pz >= 0.0 && pz <= (double)(a.nz - 1) && py >= 0.0 && py <= (double)(a.ny - 1) && px >= 0.0 && px <= (double)(a.nx - 1)) {
    const double cc[3] = {pz, py, px};
    const int dims[3] = {a.nz, a.ny, a.nx};
    double w[3][4];
    int id[3][4];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const double f = floor(cc[d]);
      const double y = cc[d] - f, zc = 1.0 - y;
      w[d][1] = (y * y * (y - 2.0) * 3.0 + 4.0) / 6.0;
      w[d][2] = (zc * zc * (zc - 2.0) * 3.0 + 4.0) / 6.0;
      w[d][0] = zc * zc * zc / 6.0;
      w[d][3] = 1.0 - w[d][0] - w[d][1] - w[d][2];
      const int start = (int)f - 1;
#pragma unroll
      for (int q = 0; q < 4; ++q) id[d][q] = spline_mirror(start + q, dims[d]);
    }
    double acc = 0.0;
    for (int q0 = 0; q0 < 4; ++q0)
      for (int q1 = 0; q1 < 4; ++q1) {
        const double* const row = c + ((int64_t)id[0][q0] * a.ny + id[1][q1]) * a.nx;
        const double w01 = w[0][q0] * w[1][q1];
#pragma unroll
        for (int q2 = 0; q2 < 4; ++q2) acc += row[id[2][q2]] * (w01 * w[2][q2]);
      }
    v = (float)acc;
  }
  out[t] = v;
}

__global__ void k_f32_to_f64(const float* __restrict__ in, int64_t n, double* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) out[t] = (double)in[t];
}
}  // namespace

extern "C" {

int hh_affine_transform_2d(int device, const float* data, int ny, int nx, const double matrix[4], const double offset[2],
                           float* out) try {
  if (!data || !out || !matrix || !offset || ny < 1 || nx < 1)
    return fail(nullptr, HH_ERR_ARG, "hh_affine_transform_2d: bad argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
    return fail(nullptr, HH_ERR_HIP, "hh_affine_transform_2d: no such HIP device (there is no CPU fallback)");
  const size_t bytes = (size_t)ny * nx * sizeof(float);
  float *d_in = nullptr, *d_out = nullptr;
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = hipMalloc(&d_in, bytes);
  if (e == hipSuccess) e = hipMalloc(&d_out, bytes);
  if (e == hipSuccess) e = hipMemcpy(d_in, data, bytes, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_affine_bilinear, dim3((nx + 255) / 256, ny), dim3(256), 0, 0, d_in, ny, nx, matrix[0], matrix[1],
                       matrix[2], matrix[3], offset[0], offset[1], d_out);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(out, d_out, bytes, hipMemcpyDeviceToHost);
  (void)hipFree(d_in);
  (void)hipFree(d_out);
  if (e != hipSuccess) return fail(nullptr, HH_ERR_HIP, std::string("hh_affine_transform_2d: ") + hipGetErrorString(e));
  return HH_OK;
} HH_CATCH_CTX(nullptr, "hh_affine_transform_2d")

int hh_transform_map(int device, const float* data, const int32_t shape[3], double scale, double rot_degree, double tilt_degree,
                     double psi_degree, double dx, double dy, double dz, float* out) try {
  if (!data || !out || !shape || shape[0] < 1 || shape[1] < 1 || shape[2] < 1)
    return fail(nullptr, HH_ERR_ARG, "hh_transform_map: bad argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
    return fail(nullptr, HH_ERR_HIP, "hh_transform_map: no such HIP device (there is no CPU fallback)");
  const int nz = shape[0], ny = shape[1], nx = shape[2];
  const int64_t total = (int64_t)nz * ny * nx;
  MapArgs a{};
  a.nz = nz; a.ny = ny; a.nx = nx;
  a.scale = scale;
  a.ox = (double)(nx / 2) - dx;
  a.oy = (double)(ny / 2) - dy;
  a.oz = (double)(nz / 2) - dz;
  {  // intrinsic "ZYZ" Euler angles (scipy Rotation.from_euler("ZYZ", (rot, tilt, psi))): Rz(rot) Ry(tilt) Rz(psi)
    const double d2r = M_PI / 180.0;
    const double c1 = std::cos(rot_degree * d2r), s1 = std::sin(rot_degree * d2r), c2 = std::cos(tilt_degree * d2r),
                 s2 = std::sin(tilt_degree * d2r), c3 = std::cos(psi_degree * d2r), s3 = std::sin(psi_degree * d2r);
    const double rz1[9] = {c1, -s1, 0, s1, c1, 0, 0, 0, 1}, ry[9] = {c2, 0, s2, 0, 1, 0, -s2, 0, c2}, rz3[9] = {c3, -s3, 0, s3, c3, 0, 0, 0, 1};
    double t[9];
    for (int r = 0; r < 3; ++r)
      for (int cidx = 0; cidx < 3; ++cidx) t[3 * r + cidx] = rz1[3 * r] * ry[cidx] + rz1[3 * r + 1] * ry[3 + cidx] + rz1[3 * r + 2] * ry[6 + cidx];
    for (int r = 0; r < 3; ++r)
      for (int cidx = 0; cidx < 3; ++cidx) a.m[3 * r + cidx] = t[3 * r] * rz3[cidx] + t[3 * r + 1] * rz3[3 + cidx] + t[3 * r + 2] * rz3[6 + cidx];
  }
  float *d_in = nullptr, *d_out = nullptr;
  double* d_c = nullptr;
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = hipMalloc(&d_in, (size_t)total * sizeof(float));
  if (e == hipSuccess) e = hipMalloc(&d_out, (size_t)total * sizeof(float));
  if (e == hipSuccess) e = hipMalloc(&d_c, (size_t)total * sizeof(double));
  if (e == hipSuccess) e = hipMemcpy(d_in, data, (size_t)total * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    const unsigned blocks = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(k_f32_to_f64, dim3(blocks), dim3(256), 0, 0, d_in, total, d_c);
    // axis 0 (z), 1 (y), 2 (x), in SciPy's order: lines = the product of the other two sides
    const int len[3] = {nz, ny, nx};
    const int64_t inner[3] = {(int64_t)ny * nx, (int64_t)nx, 1};
    for (int ax = 0; ax < 3; ++ax) {
      const int64_t lines = total / len[ax];
      hipLaunchKernelGGL(k_spline_prefilter, dim3((unsigned)((lines + 127) / 128)), dim3(128), 0, 0, d_c, lines, len[ax], inner[ax],
                         inner[ax]);
    }
    hipLaunchKernelGGL(k_map_cubic, dim3(blocks), dim3(256), 0, 0, d_c, a, d_out);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(out, d_out, (size_t)total * sizeof(float), hipMemcpyDeviceToHost);
  (void)hipFree(d_in);
  (void)hipFree(d_out);
  (void)hipFree(d_c);
  if (e != hipSuccess) return fail(nullptr, HH_ERR_HIP, std::string("hh_transform_map: ") + hipGetErrorString(e));
  return HH_OK;
} HH_CATCH_CTX(nullptr, "hh_transform_map")

int hh_apply_helical_symmetry(int device, const float* data, const int32_t in_shape[3], double apix,
                              double twist_degree, double rise_angstrom, int csym, double fraction,
                              const int32_t new_size[3], double new_apix, float* out, int32_t out_shape[3],
                              double* kernel_ms) try {
  if (!data || !in_shape || !new_size || !out_shape)
    return fail(nullptr, HH_ERR_ARG, "hh_apply_helical_symmetry: NULL argument");
  const int nz0 = in_shape[0], ny0 = in_shape[1], nx0 = in_shape[2];
  const int nz1 = new_size[0], ny1 = new_size[1], nx1 = new_size[2];
  if (nz0 < 2 || ny0 < 2 || nx0 < 2 || nz1 < 1 || ny1 < 1 || nx1 < 1 || csym < 1 || !(apix > 0) || !(new_apix > 0) ||
      !(rise_angstrom > 0))
    return fail(nullptr, HH_ERR_ARG, "hh_apply_helical_symmetry: bad shape or parameter");
  SymArgs a{};
  a.nz0 = nz0; a.ny0 = ny0; a.nx0 = nx0;
  const bool same = nz0 == nz1 && ny0 == ny1 && nx0 == nx1;
  a.nz = std::max(nz0, nz1); a.ny = std::max(ny0, ny1); a.nx = std::max(nx0, nx1);
  if (same || (a.nz == nz1 && a.ny == ny1 && a.nx == nx1)) {  // transforms.py:158: no crop
    a.cz = a.cy = a.cx = 0;
    a.oz = a.nz; a.oy = a.ny; a.ox = a.nx;
  } else {  // Python slice [n//2 - n1//2 : n//2 + n1//2] on every axis
    a.cz = std::max(0, a.nz / 2 - nz1 / 2); a.oz = std::min(a.nz, a.nz / 2 + nz1 / 2) - a.cz;
    a.cy = std::max(0, a.ny / 2 - ny1 / 2); a.oy = std::min(a.ny, a.ny / 2 + ny1 / 2) - a.cy;
    a.cx = std::max(0, a.nx / 2 - nx1 / 2); a.ox = std::min(a.nx, a.nx / 2 + nx1 / 2) - a.cx;
  }
  out_shape[0] = a.oz; out_shape[1] = a.oy; out_shape[2] = a.ox;
  if (!out) return HH_OK;  // shape query
  a.apix = apix; a.new_apix = new_apix; a.rise = rise_angstrom; a.csym = csym;
  a.hmax = std::max(1, (int)((double)a.nz * new_apix / rise_angstrom));  // transforms.py:88
  // z range of the input that carries density (transforms.py:92-99)
  std::vector<double> prof(nz0, 0.0);
  for (int k = 0; k < nz0; ++k) {
    double acc = 0;
    const float* p = data + (size_t)k * ny0 * nx0;
    for (size_t q = 0; q < (size_t)ny0 * nx0; ++q) acc += p[q];
    prof[k] = acc;
  }
  const double thr = 0.01 * *std::max_element(prof.begin(), prof.end());
  int z0 = -1, z1 = -1;
  for (int k = 0; k < nz0; ++k)
    if (prof[k] > thr) { if (z0 < 0) z0 = k; z1 = k; }
  if (z0 < 0) return fail(nullptr, HH_ERR_ARG, "hh_apply_helical_symmetry: the volume has no density above 1 % of its peak slice");
  const int zmid = (z0 + z1) / 2 + (z0 + z1) % 2;
  const int half = (int)((double)nz0 * fraction + 0.5) / 2;
  a.z0 = std::max(z0, zmid - half);
  a.z1 = std::min(z1, zmid + half);
  std::vector<double> rot((size_t)(2 * a.hmax + 1) * csym * 2);
  for (int hi = -a.hmax; hi <= a.hmax; ++hi)
    for (int ci = 0; ci < csym; ++ci) {
      const double r = (twist_degree * hi + 360.0 * ci / csym) * (M_PI / 180.0);  // np.deg2rad
      rot[2 * ((size_t)(hi + a.hmax) * csym + ci)] = std::cos(r);
      rot[2 * ((size_t)(hi + a.hmax) * csym + ci) + 1] = std::sin(r);
    }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
    return fail(nullptr, HH_ERR_HIP, "hh_apply_helical_symmetry: no such HIP device (no CPU fallback)");
  HH_HIP(nullptr, hipSetDevice(device));
  const size_t n_in = (size_t)nz0 * ny0 * nx0, n_out = (size_t)a.oz * a.oy * a.ox;
  float *d_in = nullptr, *d_out = nullptr;
  double* d_rot = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t e = hipMalloc(&d_in, n_in * sizeof(float));
  if (e == hipSuccess) e = hipMalloc(&d_out, n_out * sizeof(float));
  if (e == hipSuccess) e = hipMalloc(&d_rot, rot.size() * sizeof(double));
  if (e == hipSuccess) e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  if (e == hipSuccess) e = hipMemcpy(d_in, data, n_in * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_rot, rot.data(), rot.size() * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    a.data = d_in; a.out = d_out; a.rot = d_rot;
    (void)hipEventRecord(e0, nullptr);
    hipLaunchKernelGGL(k_apply_helical_symmetry, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, nullptr, a);
    (void)hipEventRecord(e1, nullptr);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(out, d_out, n_out * sizeof(float), hipMemcpyDeviceToHost);
  if (e == hipSuccess && kernel_ms) {
    float ms = 0.f;
    e = hipEventElapsedTime(&ms, e0, e1);
    *kernel_ms = ms;
  }
  (void)hipFree(d_in); (void)hipFree(d_out); (void)hipFree(d_rot);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (e != hipSuccess) return fail(nullptr, HH_ERR_HIP, std::string("hh_apply_helical_symmetry: ") + hipGetErrorString(e));
  return HH_OK;
} HH_CATCH_CTX(nullptr, "hh_apply_helical_symmetry")

int hh_calibrate_traffic(hh_ctx* c, int mode, int64_t bytes) try {
  if (!c || bytes <= 0 || (mode != 0 && mode != 1)) return fail(c, HH_ERR_ARG, "hh_calibrate_traffic: bad argument");
  if (c->general) return fail(c, HH_ERR_ARG, "hh_calibrate_traffic: square power-of-two contexts only");
  HH_HIP(c, hipSetDevice(c->device));
  const size_t unit = (size_t)256 * 512 * sizeof(float2);  // one 512-side half spectrum (1 MiB)
  const size_t n_units = std::max<size_t>(1, (size_t)bytes / unit);
  float2* buf = nullptr;
  float* sink = nullptr;
  HH_HIP(c, hipMalloc(&buf, n_units * unit));
  HH_HIP(c, hipMalloc(&sink, sizeof(float)));
  hipError_t e = hipMemsetAsync(buf, 0, n_units * unit, c->stream);
  if (e == hipSuccess) {
    if (mode == 0)
      hipLaunchKernelGGL(k_calib_read, dim3(2048), dim3(512), 0, c->stream, reinterpret_cast<const float4*>(buf),
                         n_units * 32, sink);
    else
      hipLaunchKernelGGL(k_calib_write, dim3(2048), dim3(512), 0, c->stream, buf, n_units);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(buf);
  (void)hipFree(sink);
  if (e != hipSuccess) return fail(c, HH_ERR_HIP, std::string("hh_calibrate_traffic: ") + hipGetErrorString(e));
  return HH_OK;
} HH_CATCH_CTX(c, "hh_calibrate_traffic")

int hh_profile_enable(hh_ctx* c, int on) try {
  if (!c) return HH_ERR_ARG;
  c->profiling = on > 0 ? on : 0;
  return HH_OK;
} HH_CATCH_CTX(c, "hh_profile_enable")

int hh_profile_reset(hh_ctx* c) try {
  if (!c) return HH_ERR_ARG;
  HH_HIP(c, hipSetDevice(c->device));
  HH_HIP(c, hipStreamSynchronize(c->stream));
  c->events_used = 0;
  c->prof_candidates = 0;
  return HH_OK;
} HH_CATCH_CTX(nullptr, "hh_profile_reset")

int hh_profile_get(hh_ctx* c, hh_profile* out) try {
  if (!c || !out) return HH_ERR_ARG;
  HH_HIP(c, hipSetDevice(c->device));
  HH_HIP(c, hipStreamSynchronize(c->stream));
  hh_profile p{};
  for (size_t i = 0; i < c->events_used; ++i) {
    float ms = 0.f;
    HH_HIP(c, hipEventElapsedTime(&ms, c->events[i].a, c->events[i].b));
    switch (c->events[i].kind) {
      case 0: p.ms_first_pass += ms; p.n_first_pass++; break;
      case 1: p.ms_second_pass += ms; p.n_second_pass++; break;
      case 3: p.ms_centres += ms; p.n_centres++; break;
      default: p.ms_finalize += ms; p.n_finalize++; break;
    }
  }
  p.candidates = c->prof_candidates;
  *out = p;
  return HH_OK;
} HH_CATCH_CTX(c, "hh_profile_get")

}  // extern "C"

#include "image_prep.inc"    // pre-sweep image preparation that is scikit-image in the reference (warp, rescale, closing + moments)
#include "fourier_zoom.inc"  // compute_power_spectra with cutoff_res / output_size: direct non-uniform DFT (hh_power_spectrum_zoom)
#include "path_a_host.inc"  // Path A: host side and C ABI (hh_pa_*)
#include "path_a_batch.inc"  // Path A for many candidates at once, device-resident solve (hh_pab_*)
