// gen_rows.hip — the general-size sweep's row kernel for nx = R1 * R2 (R1, R2 <= 32): second translation unit of
// libhelicon_hip.so (declared in gen_rows.h, driven by general_host.inc; k_gen_fused in general_sizes.inc stays the
// kernel for every other nx).
//
// What the reference does for such an image is the same as for any other (utils.py:31-47, 91-106: rfft2 of the
// simulated projection, log1p|F|, masked Pearson score); the algorithm is general_sizes.inc's — a candidate's row of the
// column-transformed lattice image H[ky][x] is rebuilt from its run's table slice and its column factors, transformed
// along x, and the three masked moments of the row go to k_finalize.  What differs is how the row is transformed:
//
//   * one wavefront carries RPW = 64 / max(R1, R2) rows at once, each on L = max(R1, R2) adjacent lanes; a workgroup is
//     four wavefronts (RPB = 4 RPW rows of one ky block);
//   * step 1: lane j < R2 of a row takes the R1 points x = j + R2 r, does the R1-point transform in registers
//     (compile-time twiddles, no index arithmetic), multiplies by W_nx^(j k1) (twiddles held in registers for all
//     candidates) and writes A[j][k1] transposed with an odd stride; step 2: lane k1 < R1 takes A[.][k1], does the
//     R2-point transform in registers and holds the bins kx = k1 + R1 k2 — whose weights it has kept in registers since
//     the workgroup started.  The row crosses LDS once between the steps and once after the build, in place, and
//     because a row never leaves its wavefront the steps need no barrier (LDS instructions of a wavefront execute in
//     order);
//   * the candidate's column factors are fetched into registers while the previous candidate is transformed and
//     parked in the other half of a double LDS buffer: one workgroup barrier per candidate.
//
// k_gen_fused needs 877 vector + 132 LDS instructions per (row, candidate) at nx = 400 (four Stockham stages of radix
// 4 / 5 with run-time index arithmetic, 22 % of the lanes idle); this kernel's counts are in DESIGN.md.
#include "gen_rows.h"

#include <algorithm>
#include <cstdint>
#include <type_traits>

namespace {

// ---- compile-time twiddles ---------------------------------------------------------------------------------------
constexpr double GR_PI = 3.14159265358979323846264338327950288;

constexpr double gr_sin_small(double x) {   // |x| <= pi / 4
  const double x2 = x * x;
  double term = x, sum = x;
  for (int i = 1; i < 14; ++i) {
    term *= -x2 / (double)((2 * i) * (2 * i + 1));
    sum += term;
  }
  return sum;
}
constexpr double gr_cos_small(double x) {
  const double x2 = x * x;
  double term = 1.0, sum = 1.0;
  for (int i = 1; i < 14; ++i) {
    term *= -x2 / (double)((2 * i - 1) * (2 * i));
    sum += term;
  }
  return sum;
}
// cos(2 pi a / b) by octant symmetry (exact zeros and ones at the multiples of b / 4)
constexpr double gr_cos_frac(long long a, long long b) {
  a = ((a % b) + b) % b;
  if (2 * a > b) a = b - a;                                   // cos is even about pi
  if (4 * a > b) {                                            // second quadrant: -cos(pi - t)
    const long long a2 = b - 2 * a, b2 = 2 * b;               // (1/2 - a/b) = (b - 2a) / (2b) <= 1/4
    if (8 * a2 > b2) return -gr_sin_small(2.0 * GR_PI * (double)(b2 - 4 * a2) / (double)(4 * b2));
    return -gr_cos_small(2.0 * GR_PI * (double)a2 / (double)b2);
  }
  if (8 * a > b) return gr_sin_small(2.0 * GR_PI * (double)(b - 4 * a) / (double)(4 * b));
  return gr_cos_small(2.0 * GR_PI * (double)a / (double)b);
}
constexpr double gr_sin_frac(long long a, long long b) { return gr_cos_frac(4 * a - b, 4 * b); }   // sin t = cos(t - pi/2)

template <int I, int N, class F>
__device__ __forceinline__ void gr_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    gr_for<I + 1, N>(f);
  }
}

__device__ __forceinline__ float2 gr_add(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 gr_sub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 gr_mul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// v * exp(-2 pi i E / R) for compile-time E, R
template <int E0, int R>
__device__ __forceinline__ float2 gr_twiddle(float2 v) {
  constexpr int E = ((E0 % R) + R) % R;
  if constexpr (E == 0) {
    return v;
  } else if constexpr (4 * E == R) {
    return make_float2(v.y, -v.x);          // -i
  } else if constexpr (2 * E == R) {
    return make_float2(-v.x, -v.y);
  } else if constexpr (4 * E == 3 * R) {
    return make_float2(-v.y, v.x);          // +i
  } else if constexpr (8 * E == R) {
    constexpr float h = 0.70710678118654752440f;
    return make_float2((v.x + v.y) * h, (v.y - v.x) * h);
  } else if constexpr (8 * E == 3 * R) {
    constexpr float h = 0.70710678118654752440f;
    return make_float2((v.y - v.x) * h, -(v.x + v.y) * h);
  } else {
    constexpr float c = (float)gr_cos_frac(E, R), s = (float)-gr_sin_frac(E, R);   // W = c + i s
    return make_float2(v.x * c - v.y * s, v.x * s + v.y * c);
  }
}

constexpr int gr_first_factor(int r) {
  if (r <= 5 || r == 8) return r;
  if (r % 4 == 0) return 4;
  if (r % 2 == 0) return 2;
  for (int p = 3; p * p <= r; p += 2)
    if (r % p == 0) return p;
  return r;
}

// ---- in-register transforms ----------------------------------------------------------------------------------------
template <int R>
__device__ __forceinline__ void gr_dft(float2 (&a)[R]);

template <>
__device__ __forceinline__ void gr_dft<2>(float2 (&a)[2]) {
  const float2 t = a[0];
  a[0] = gr_add(t, a[1]);
  a[1] = gr_sub(t, a[1]);
}
template <>
__device__ __forceinline__ void gr_dft<3>(float2 (&a)[3]) {
  constexpr float s = 0.86602540378443864676f;
  const float2 t1 = gr_add(a[1], a[2]), t2 = gr_sub(a[1], a[2]);
  const float2 m = make_float2(a[0].x - 0.5f * t1.x, a[0].y - 0.5f * t1.y);
  const float2 r = make_float2(s * t2.y, -s * t2.x);
  a[0] = gr_add(a[0], t1);
  a[1] = gr_add(m, r);
  a[2] = gr_sub(m, r);
}
template <>
__device__ __forceinline__ void gr_dft<4>(float2 (&a)[4]) {
  const float2 s0 = gr_add(a[0], a[2]), d0 = gr_sub(a[0], a[2]);
  const float2 s1 = gr_add(a[1], a[3]), t = gr_sub(a[1], a[3]);
  const float2 d1 = make_float2(t.y, -t.x);
  a[0] = gr_add(s0, s1);
  a[2] = gr_sub(s0, s1);
  a[1] = gr_add(d0, d1);
  a[3] = gr_sub(d0, d1);
}
template <>
__device__ __forceinline__ void gr_dft<5>(float2 (&a)[5]) {
  constexpr float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
  constexpr float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
  const float2 p1 = gr_add(a[1], a[4]), m1 = gr_sub(a[1], a[4]), p2 = gr_add(a[2], a[3]), m2 = gr_sub(a[2], a[3]);
  const float2 x0 = a[0];
  const float2 u1 = make_float2(x0.x + c1 * p1.x + c2 * p2.x, x0.y + c1 * p1.y + c2 * p2.y);
  const float2 u2 = make_float2(x0.x + c2 * p1.x + c1 * p2.x, x0.y + c2 * p1.y + c1 * p2.y);
  const float2 v1 = make_float2(s1 * m1.y + s2 * m2.y, -(s1 * m1.x + s2 * m2.x));
  const float2 v2 = make_float2(s2 * m1.y - s1 * m2.y, -(s2 * m1.x - s1 * m2.x));
  a[0] = make_float2(x0.x + p1.x + p2.x, x0.y + p1.y + p2.y);
  a[1] = gr_add(u1, v1);
  a[4] = gr_sub(u1, v1);
  a[2] = gr_add(u2, v2);
  a[3] = gr_sub(u2, v2);
}
template <>
__device__ __forceinline__ void gr_dft<8>(float2 (&a)[8]) {
  constexpr float h = 0.70710678118654752440f;
  float2 e[4] = {gr_add(a[0], a[4]), gr_add(a[1], a[5]), gr_add(a[2], a[6]), gr_add(a[3], a[7])};
  const float2 o0 = gr_sub(a[0], a[4]), t1 = gr_sub(a[1], a[5]), t2 = gr_sub(a[2], a[6]), t3 = gr_sub(a[3], a[7]);
  float2 o[4] = {o0, make_float2((t1.x + t1.y) * h, (t1.y - t1.x) * h), make_float2(t2.y, -t2.x),
                 make_float2((t3.y - t3.x) * h, -(t3.x + t3.y) * h)};
  gr_dft<4>(e);
  gr_dft<4>(o);
  a[0] = e[0]; a[1] = o[0]; a[2] = e[1]; a[3] = o[1];
  a[4] = e[2]; a[5] = o[2]; a[6] = e[3]; a[7] = o[3];
}

// an odd prime p: X[m] = a0 + sum_{r=1}^{(p-1)/2} [ (a_r + a_{p-r}) cos(2 pi r m / p) - i (a_r - a_{p-r}) sin(2 pi r m / p) ]
template <int P>
__device__ __forceinline__ void gr_dft_prime(float2 (&a)[P]) {
  constexpr int H = (P - 1) / 2;
  float2 sp[H], sm[H];
  gr_for<0, H>([&](auto rc) {
    constexpr int r = decltype(rc)::value + 1;
    sp[r - 1] = gr_add(a[r], a[P - r]);
    sm[r - 1] = gr_sub(a[r], a[P - r]);
  });
  const float2 x0 = a[0];
  float2 tot = x0;
  gr_for<0, H>([&](auto rc) { tot = gr_add(tot, sp[decltype(rc)::value]); });
  a[0] = tot;
  gr_for<0, H>([&](auto mc) {
    constexpr int m = decltype(mc)::value + 1;
    float2 u = x0, v = make_float2(0.f, 0.f);   // u: cosine part, v = sum (a_r - a_{p-r}) sin
    gr_for<0, H>([&](auto rc) {
      constexpr int r = decltype(rc)::value + 1;
      constexpr float c = (float)gr_cos_frac((long long)r * m, P), s = (float)gr_sin_frac((long long)r * m, P);
      u.x = fmaf(c, sp[r - 1].x, u.x);
      u.y = fmaf(c, sp[r - 1].y, u.y);
      v.x = fmaf(s, sm[r - 1].x, v.x);
      v.y = fmaf(s, sm[r - 1].y, v.y);
    });
    // -i v = (v.y, -v.x)
    a[m] = make_float2(u.x + v.y, u.y - v.x);
    a[P - m] = make_float2(u.x - v.y, u.y + v.x);
  });
}

template <int R>
__device__ __forceinline__ void gr_dft(float2 (&a)[R]) {
  constexpr int P = gr_first_factor(R);
  if constexpr (P == R) {
    gr_dft_prime<R>(a);
  } else {
    // R = P Q, x = q + Q p, k = k1 + P k2:  X[k1 + P k2] = sum_q W_Q^(q k2) [ W_R^(q k1) sum_p W_P^(p k1) a[q + Q p] ]
    constexpr int Q = R / P;
    float2 b[R];   // b[k1 * Q + q]
    gr_for<0, Q>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      float2 t[P];
      gr_for<0, P>([&](auto pc) { t[decltype(pc)::value] = a[q + Q * decltype(pc)::value]; });
      gr_dft<P>(t);
      gr_for<0, P>([&](auto kc) {
        constexpr int k1 = decltype(kc)::value;
        b[k1 * Q + q] = gr_twiddle<q * k1, R>(t[k1]);
      });
    });
    gr_for<0, P>([&](auto kc) {
      constexpr int k1 = decltype(kc)::value;
      float2 t[Q];
      gr_for<0, Q>([&](auto qc) { t[decltype(qc)::value] = b[k1 * Q + decltype(qc)::value]; });
      gr_dft<Q>(t);
      gr_for<0, Q>([&](auto k2c) { a[k1 + P * decltype(k2c)::value] = t[decltype(k2c)::value]; });
    });
  }
}

// ---- shape of one instantiation -------------------------------------------------------------------------------------
#ifndef GR_ABLATE
#define GR_ABLATE 0   // timing-only builds (tools/variants): 1 no build, 2 no step 1, 4 no step 2, 8 no moments, 16 no reduction
#endif
#ifndef GR_DUAL
#define GR_DUAL 0     // 1: two rows per lane slot where step 1 leaves half of a slot's lanes idle (GrShape::DUAL) — correct (all 109 sizes against the oracle) and slower where it matters: 200 x 200 16.4 -> 14.7 M candidates/s, 96 x 128 46.9 -> 40.9 M, 250 x 250 9.66 -> 10.04 M (two workgroups per compute unit instead of three)
#endif
constexpr int GR_WAVES = 4;
constexpr int GR_THREADS = 64 * GR_WAVES;
constexpr int GR_KG_PF = 16;     // rows of column factors the register prefetch is sized for (more: k_gen_fused)

constexpr int gr_row_len(int r1, int r2) {
  const int l = r1 > r2 ? r1 : r2, s = r1 | 1, nx = r1 * r2, nxp = (nx + 3) / 4 * 4;
  int len = r2 * s > nxp ? r2 * s : nxp;
  const int want = (l + (l & 1)) % 32;   // rows follow each other in bank space like the lanes that hold them
  while (len % 32 != want) ++len;
  return len;
}

template <int R1, int R2>
struct GrShape {
  static constexpr int L = R1 > R2 ? R1 : R2;       // lanes per row slot
  static constexpr int RPW = 64 / L;                // row slots per wavefront
  // Step 1 keeps only R2 of a slot's L lanes busy.  When two sets of R2 lanes fit a slot (2 R2 <= L), a wavefront carries TWO
  // rows per slot: step 1 transforms both at once (lanes [0, R2) the first row's columns, [R2, 2 R2) the second's), step 2 and
  // the moments run once per row on all R1 lanes (200 = 20 x 10: the 20-point transforms of step 1 ran on 30 of 64 lanes).
  // (short rows only: above 256 points the second row's accumulators and weights do not fit the registers — 48 to 160 bytes
  // of scratch at 28 x 14 and 32 x 16 when tried)
  static constexpr bool DUAL = GR_DUAL && 2 * R2 <= L && R1 >= R2 && R1 * R2 <= 256;
  static constexpr int PH = DUAL ? 2 : 1;           // rows per slot
  static constexpr int RW = RPW * PH;               // rows per wavefront (row ph * RPW + slot)
  static constexpr int RPB = RW * GR_WAVES;         // rows per workgroup
  static constexpr int S = R1 | 1;                  // odd stride of the transposed layout between the steps
  static constexpr int NX = R1 * R2;
  static constexpr int NXP = (NX + 3) / 4 * 4;
  static constexpr int NG = NXP / 4;                // column groups of four
  static constexpr int CGS = NG + 4;                // ints per candidate in cgs
  static constexpr int CGSP = (CGS + 3) / 4 * 4;
  static constexpr int ROWLEN = gr_row_len(R1, R2);
  static constexpr int PFC = (CGS + GR_THREADS - 1) / GR_THREADS;
  static constexpr int PF = (GR_KG_PF * NG + GR_THREADS - 1) / GR_THREADS;   // float4 of column factors one thread holds in flight
  // Rows above ~256 points leave room for two workgroups per CU (LDS), so each wavefront may use 256 registers and keeps
  // its step-1 twiddles in them; shorter rows fit three workgroups and read the twiddles from an LDS table instead.
  // (the widest radices leave no registers for either table: twiddles and weights then come from global memory / L1)
  static constexpr bool TW_REGS = NX > 256 && R1 + R2 <= 48;
  static constexpr bool TW_LDS = NX <= 256;
  static constexpr bool W_REGS = R1 + R2 <= 48;
  static constexpr int MIN_BLOCKS = (NX > 256 || DUAL) ? 2 : 3;   // (DUAL: twice the rows in LDS, two workgroups per compute unit at most)
  static_assert(RPW >= 1 && L <= 64, "radix too large");
};

__device__ __forceinline__ void gr_wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int R1, int R2>
__global__ __launch_bounds__(GR_THREADS, (GrShape<R1, R2>::MIN_BLOCKS)) void k_gen_rows(GenRowsArgs a) {
  using C = GrShape<R1, R2>;
  extern __shared__ __attribute__((aligned(16))) unsigned char gr_smem[];
  float2* const rowsb = reinterpret_cast<float2*>(gr_smem);                          // [RPB][ROWLEN]
  float2* const gs = rowsb + (size_t)C::RPB * C::ROWLEN;                             // [RPB][rows_lds]
  float* const egb = reinterpret_cast<float*>(gs + (size_t)C::RPB * a.rows_lds);     // [halves][kg][NXP]
  int* const cgb = reinterpret_cast<int*>(egb + (size_t)a.halves * a.kg * C::NXP);   // [halves][CGSP]
  float2* const twl = reinterpret_cast<float2*>(cgb + a.halves * C::CGSP);           // [R1][R2] W_nx^(j k1) (short rows only)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int m_raw = lane / C::L, j = lane - m_raw * C::L;
  const bool lane_ok = m_raw < C::RPW;
  const int m = lane_ok ? m_raw : C::RPW - 1;
  // the lane's rows inside the workgroup: slot m holds row ph * RPW + m of the wavefront for ph < PH
  const int rib0 = wave * C::RW + m;
  const int run = a.layer_run[blockIdx.y];
  const int cfirst = a.layer_first[blockIdx.y], nc = a.layer_count[blockIdx.y];
  const int n_e4 = a.kg * C::NG;

  // the run's table slice for the workgroup's rows
  {
    const int rows = (2 * a.run_imax[run] + 1) * a.n_units;
    const float2* const tab = a.table + (size_t)run * a.cap * a.nky;
    const int ky0 = blockIdx.x * C::RPB;
    for (int e = tid; e < a.rows_lds * C::RPB; e += GR_THREADS) {
      const int cr = e / C::RPB, r = e - cr * C::RPB;
      gs[(size_t)r * a.rows_lds + cr] = (cr < rows && ky0 + r < a.nky) ? tab[(size_t)cr * a.nky + ky0 + r] : make_float2(0.f, 0.f);
    }
  }
  // step-1 twiddles W_nx^(j k1) and the weights of the bins kx = j + R1 k2 this lane will hold after step 2
  // step 1: (DUAL) lanes [R2, 2 R2) of a slot work on the slot's second row
  const int ph1 = C::DUAL && j >= R2 ? 1 : 0;
  const int j1 = min(j - ph1 * R2, R2 - 1), j2 = min(j, R1 - 1);
  float2 tw1[C::TW_REGS ? R1 : 1];
  if constexpr (C::TW_REGS) {
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) tw1[k1] = a.tw_nx[j1 * k1];
  } else if constexpr (C::TW_LDS) {
    for (int e = tid; e < C::NX; e += GR_THREADS) {
      const int k1 = e / R2, jj = e - k1 * R2;
      twl[e] = a.tw_nx[jj * k1];
    }
  }
  bool w_ok[C::PH];
  const float2* wrow[C::PH];
  float2 wreg[C::PH][C::W_REGS ? R2 : 1];
#pragma unroll
  for (int ph = 0; ph < C::PH; ++ph) {
    const int row = blockIdx.x * C::RPB + rib0 + ph * C::RPW;
    w_ok[ph] = lane_ok && row < a.nky && j < R1;
    wrow[ph] = a.w2 + (size_t)(w_ok[ph] ? row : 0) * C::NX + j2;
    if constexpr (C::W_REGS) {
#pragma unroll
      for (int k2 = 0; k2 < R2; ++k2) wreg[ph][k2] = w_ok[ph] ? wrow[ph][R1 * k2] : make_float2(0.f, 0.f);
    }
  }

  // column factors of the first candidate
  float4 pf[C::PF];
  int pc[C::PFC];
#define GR_FETCH(CAND)                                                                                        \
  {                                                                                                           \
    const float4* const src_ = reinterpret_cast<const float4*>(a.eg + (size_t)(CAND) * a.kg * C::NXP);        \
    _Pragma("unroll") for (int i = 0; i < C::PF; ++i)                                                         \
      pf[i] = tid + GR_THREADS * i < n_e4 ? src_[tid + GR_THREADS * i] : make_float4(0.f, 0.f, 0.f, 0.f);     \
    _Pragma("unroll") for (int i = 0; i < C::PFC; ++i)                                                        \
      pc[i] = tid + GR_THREADS * i < C::CGS ? a.cgs[(size_t)(CAND) * C::CGS + tid + GR_THREADS * i] : 0;      \
  }
#define GR_PARK(HALF)                                                                                         \
  {                                                                                                           \
    float4* const dst_ = reinterpret_cast<float4*>(egb + (size_t)(HALF) * a.kg * C::NXP);                     \
    _Pragma("unroll") for (int i = 0; i < C::PF; ++i)                                                         \
      if (tid + GR_THREADS * i < n_e4) dst_[tid + GR_THREADS * i] = pf[i];                                    \
    _Pragma("unroll") for (int i = 0; i < C::PFC; ++i)                                                        \
      if (tid + GR_THREADS * i < C::CGS) cgb[(HALF) * C::CGSP + tid + GR_THREADS * i] = pc[i];                \
  }
  if (nc > 0) {
    GR_FETCH(cfirst);
    GR_PARK(0);
  }
  __syncthreads();

#pragma unroll 1
  for (int cc = 0; cc < nc; ++cc) {
    const int half = a.halves == 2 ? (cc & 1) : 0;
    if (cc + 1 < nc) GR_FETCH(cfirst + cc + 1);
    const float* const eg = egb + (size_t)half * a.kg * C::NXP;
    const int* const cg = cgb + half * C::CGSP;
    const int kc = max(0, min(a.kg, cg[C::NG]));
    // ---- the wavefront's RPW rows of H, four columns per lane, natural order
    // (tried: a lane's column groups built together with the next table row's operands requested ahead — more registers
    // and moves than the waiting it saves; the build is its FMAs)
    // Passes of 64 column groups: a lane builds its group for all RPW rows (the factors are read once).  When dealing
    // the groups of the last, partial pass out as (row, group) items takes fewer item passes than there are rows per
    // wavefront (36 left-over groups x 3 rows at nx = 400: two item passes of 8 FMAs per table row instead of one
    // grouped pass of 24 with 28 lanes idle), they are.
    constexpr int REM0 = C::NG % 64, IPASS = (REM0 * C::RW + 63) / 64;
    constexpr bool FLAT = REM0 > 0 && IPASS < C::RW;
    constexpr int FULL = FLAT ? C::NG / 64 : (C::NG + 63) / 64, REM = FLAT ? REM0 : 0, ITEMS = REM * C::RW;
    if (!(GR_ABLATE & 1)) {
#pragma unroll
      for (int f = 0; f < FULL; ++f) {
        const int xg = lane + 64 * f;
        if (xg >= C::NG) break;   // (the partial pass, when it is kept grouped)
        const float2* const g0 = gs + (size_t)(wave * C::RW) * a.rows_lds + cg[xg];
        const float* const erow = eg + 4 * xg;
        float2 p[C::RW][4];
#pragma unroll
        for (int r = 0; r < C::RW; ++r)
#pragma unroll
          for (int q = 0; q < 4; ++q) p[r][q] = make_float2(0.f, 0.f);
        for (int k = 0; k < kc; ++k) {
          const float4 e4 = *reinterpret_cast<const float4*>(erow + (size_t)k * C::NXP);
#pragma unroll
          for (int r = 0; r < C::RW; ++r) {
            const float2 gk = g0[(size_t)r * a.rows_lds + k];
            p[r][0].x = fmaf(e4.x, gk.x, p[r][0].x); p[r][0].y = fmaf(e4.x, gk.y, p[r][0].y);
            p[r][1].x = fmaf(e4.y, gk.x, p[r][1].x); p[r][1].y = fmaf(e4.y, gk.y, p[r][1].y);
            p[r][2].x = fmaf(e4.z, gk.x, p[r][2].x); p[r][2].y = fmaf(e4.z, gk.y, p[r][2].y);
            p[r][3].x = fmaf(e4.w, gk.x, p[r][3].x); p[r][3].y = fmaf(e4.w, gk.y, p[r][3].y);
          }
        }
#pragma unroll
        for (int r = 0; r < C::RW; ++r) {
          float4* const dst = reinterpret_cast<float4*>(rowsb + (size_t)(wave * C::RW + r) * C::ROWLEN + 4 * xg);
          dst[0] = make_float4(p[r][0].x, p[r][0].y, p[r][1].x, p[r][1].y);
          dst[1] = make_float4(p[r][2].x, p[r][2].y, p[r][3].x, p[r][3].y);
        }
      }
      if constexpr (REM > 0) {
#pragma unroll
        for (int u = 0; u < IPASS; ++u) {
          const int it = lane + 64 * u;
          if (it < ITEMS) {
            const int r = it / REM, xg = 64 * FULL + (it - r * REM);
            const float2* const g0 = gs + (size_t)(wave * C::RW + r) * a.rows_lds + cg[xg];
            const float* const erow = eg + 4 * xg;
            float2 p0 = make_float2(0.f, 0.f), p1 = p0, p2 = p0, p3 = p0;
            for (int k = 0; k < kc; ++k) {
              const float4 e4 = *reinterpret_cast<const float4*>(erow + (size_t)k * C::NXP);
              const float2 gk = g0[k];
              p0.x = fmaf(e4.x, gk.x, p0.x); p0.y = fmaf(e4.x, gk.y, p0.y);
              p1.x = fmaf(e4.y, gk.x, p1.x); p1.y = fmaf(e4.y, gk.y, p1.y);
              p2.x = fmaf(e4.z, gk.x, p2.x); p2.y = fmaf(e4.z, gk.y, p2.y);
              p3.x = fmaf(e4.w, gk.x, p3.x); p3.y = fmaf(e4.w, gk.y, p3.y);
            }
            float4* const dst = reinterpret_cast<float4*>(rowsb + (size_t)(wave * C::RW + r) * C::ROWLEN + 4 * xg);
            dst[0] = make_float4(p0.x, p0.y, p1.x, p1.y);
            dst[1] = make_float4(p2.x, p2.y, p3.x, p3.y);
          }
        }
      }
    }
    gr_wave_fence();
    // ---- step 1: R1 points x = j + R2 r per lane
    {
      float2* const myrow = rowsb + (size_t)(rib0 + ph1 * C::RPW) * C::ROWLEN;
      float2 v[R1];
#pragma unroll
      for (int r = 0; r < R1; ++r) v[r] = myrow[j1 + R2 * r];
      if (!(GR_ABLATE & 2)) gr_dft<R1>(v);
      gr_wave_fence();   // (program order: every lane's reads above precede the writes below)
      if (lane_ok && j < C::PH * R2) {
        myrow[j1 * C::S] = v[0];
#pragma unroll
        for (int k1 = 1; k1 < R1; ++k1) {
          float2 t;
          if constexpr (C::TW_REGS) t = tw1[k1];
          else if constexpr (C::TW_LDS) t = twl[k1 * R2 + j1];
          else t = a.tw_nx[j1 * k1];
          myrow[j1 * C::S + k1] = gr_mul(v[k1], t);
        }
      }
    }
    gr_wave_fence();
    // ---- step 2 (once per row of the slot): A[.][k1 = j] -> the bins kx = j + R1 k2, and the row's three masked moments
#pragma unroll
    for (int ph = 0; ph < C::PH; ++ph) {
      const int rib = rib0 + ph * C::RPW;
      const int row = blockIdx.x * C::RPB + rib;
      const bool row_ok = lane_ok && row < a.nky;
      const float2* const myrow = rowsb + (size_t)rib * C::ROWLEN;
      float s1 = 0.f, s2 = 0.f, s3 = 0.f;
      {
        float2 v[R2];
#pragma unroll
        for (int r = 0; r < R2; ++r) v[r] = myrow[r * C::S + j2];
        if (!(GR_ABLATE & 4)) gr_dft<R2>(v);
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) {
          const float av = (GR_ABLATE & 8) ? v[k2].x + v[k2].y : __builtin_amdgcn_sqrtf(v[k2].x * v[k2].x + v[k2].y * v[k2].y);
          const float q = (GR_ABLATE & 8) ? av : a.log_flag ? __log2f(1.0f + av) : av;
          float2 w;
          if constexpr (C::W_REGS) w = wreg[ph][k2];
          else w = w_ok[ph] ? wrow[ph][R1 * k2] : make_float2(0.f, 0.f);
          // several segments: the masked q goes to HBM for the contraction with every segment's centred spectrum
          // (k_segment_corr); bins outside the mask carry no weight in any segment
          if (a.q_out && w_ok[ph]) a.q_out[(size_t)(cfirst + cc) * a.q_stride + (size_t)row * C::NX + j + R1 * k2] = w.x > 0.f ? q : 0.f;
          s1 = fmaf(w.x, q, s1);
          s2 = fmaf(w.x * q, q, s2);
          s3 = fmaf(w.y, q, s3);
        }
      }
      // sum over the row's L lanes (segments of the wavefront): lane j = 0 of each row ends with the total.  When L is a
      // multiple of 4 the rows start on quad boundaries: the first two levels are DPP quad permutes (no LDS queue, no
      // masks: every lane of a quad ends with the quad's sum), the rest shuffles across quads.
      constexpr int RED0 = (C::L % 4 == 0) ? 4 : 1;
      if constexpr (RED0 == 4) {
#define GR_QUAD_ADD(V, CTRL) V += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(V), CTRL, 0xF, 0xF, false))
        GR_QUAD_ADD(s1, 0xB1); GR_QUAD_ADD(s2, 0xB1); GR_QUAD_ADD(s3, 0xB1);   // quad_perm [1,0,3,2]
        GR_QUAD_ADD(s1, 0x4E); GR_QUAD_ADD(s2, 0x4E); GR_QUAD_ADD(s3, 0x4E);   // quad_perm [2,3,0,1]
#undef GR_QUAD_ADD
      }
#pragma unroll
      for (int off = RED0; off < ((GR_ABLATE & 16) ? 1 : C::L); off <<= 1) {
        const float t1 = __shfl_down(s1, off, 64), t2 = __shfl_down(s2, off, 64), t3 = __shfl_down(s3, off, 64);
        if (j + off < C::L) {
          s1 += t1;
          s2 += t2;
          s3 += t3;
        }
      }
      if (row_ok && j == 0) {
        double* const o = a.partials + ((size_t)(cfirst + cc) * a.nky + row) * 3;
        o[0] = s1;
        o[1] = s2;
        o[2] = s3;
      }
    }
    if (a.halves == 1) __syncthreads();   // one buffer: every wavefront has finished with this candidate's factors
    if (cc + 1 < nc) GR_PARK(a.halves == 2 ? (half ^ 1) : 0);
    __syncthreads();   // the next candidate's factors are in place
  }
}

#undef GR_FETCH
#undef GR_PARK

// ---- the instantiated factorisations -----------------------------------------------------------------------------------
// For every nx in 32 ... 1024 that is a product of two 7-smooth numbers <= 32: the pair gen_rows_plan's cost rule picks
// (R1 >= R2: the wider step runs first, so step 2 and the moments keep all of a row's lanes busy).  The list is dealt to
// four translation units (-DGEN_ROWS_PART=0..3 of this file, compiled side by side by __graft_entry__.build()); part 0
// also holds the host entry points.
#define GEN_ROWS_PAIRS_0(X) X(6, 6) X(8, 5) X(8, 7) X(9, 7) X(9, 9) X(10, 10) X(14, 7) X(14, 9) X(15, 12) \
  X(16, 8) X(16, 15) X(18, 12) X(18, 15) X(21, 7) X(21, 14) X(21, 15) X(21, 18) X(24, 21) X(25, 15) X(25, 21) \
  X(25, 25) X(27, 15) X(28, 20) X(28, 24) X(30, 24) X(32, 24) X(32, 27) X(32, 32)
#define GEN_ROWS_PAIRS_1(X) X(7, 6) X(8, 8) X(9, 5) X(10, 7) X(12, 7) X(12, 9) X(15, 7) X(15, 9) X(16, 10) \
  X(16, 12) X(16, 14) X(20, 15) X(20, 16) X(20, 18) X(24, 16) X(24, 18) X(25, 5) X(25, 7) X(27, 21) X(27, 25) \
  X(27, 27) X(28, 16) X(28, 21) X(28, 28) X(30, 15) X(30, 28) X(32, 30)
#define GEN_ROWS_PAIRS_2(X) X(8, 4) X(8, 6) X(9, 8) X(10, 5) X(10, 9) X(14, 8) X(14, 10) X(14, 12) X(14, 14) \
  X(15, 5) X(15, 15) X(18, 14) X(18, 18) X(20, 14) X(20, 20) X(21, 9) X(21, 16) X(21, 21) X(24, 20) X(24, 24) \
  X(25, 24) X(27, 18) X(28, 25) X(30, 21) X(30, 25) X(32, 25) X(32, 28)
#define GEN_ROWS_PAIRS_3(X) X(7, 5) X(7, 7) X(9, 6) X(10, 6) X(10, 8) X(12, 8) X(12, 10) X(12, 12) X(15, 10) \
  X(15, 14) X(16, 16) X(18, 9) X(18, 16) X(20, 10) X(21, 20) X(25, 10) X(25, 14) X(25, 20) X(27, 9) X(27, 20) \
  X(27, 24) X(28, 14) X(28, 27) X(30, 27) X(30, 30) X(32, 16) X(32, 20)

#ifndef GEN_ROWS_PART
#define GEN_ROWS_PART 0
#endif
#if GEN_ROWS_PART == 0
#define GR_MY_PAIRS GEN_ROWS_PAIRS_0
#elif GEN_ROWS_PART == 1
#define GR_MY_PAIRS GEN_ROWS_PAIRS_1
#elif GEN_ROWS_PART == 2
#define GR_MY_PAIRS GEN_ROWS_PAIRS_2
#else
#define GR_MY_PAIRS GEN_ROWS_PAIRS_3
#endif

#define GR_ENTRY(A, B) \
  {A, B, &k_gen_rows<A, B>, GrShape<A, B>::RPB, GrShape<A, B>::ROWLEN, GrShape<A, B>::NXP, GrShape<A, B>::PF, GrShape<A, B>::TW_LDS ? A * B : 0, \
   GrShape<A, B>::MIN_BLOCKS},
const GenRowsEntry gr_my_entries[] = {GR_MY_PAIRS(GR_ENTRY)};
#undef GR_ENTRY

}  // namespace

#define GR_PART_FN2(K) gen_rows_part_##K
#define GR_PART_FN(K) GR_PART_FN2(K)
const GenRowsEntry* GR_PART_FN(GEN_ROWS_PART)(int* n) {
  *n = (int)(sizeof(gr_my_entries) / sizeof(gr_my_entries[0]));
  return gr_my_entries;
}

#if GEN_ROWS_PART == 0
namespace {

template <class F>
void gr_each(F&& f) {
  const GenRowsEntry* (*const parts[])(int*) = {gen_rows_part_0, gen_rows_part_1, gen_rows_part_2, gen_rows_part_3};
  for (auto part : parts) {
    int n = 0;
    const GenRowsEntry* e = part(&n);
    for (int i = 0; i < n; ++i) f(e[i]);
  }
}

const GenRowsEntry* gr_find(int r1, int r2) {
  const GenRowsEntry* hit = nullptr;
  gr_each([&](const GenRowsEntry& e) { if (e.r1 == r1 && e.r2 == r2) hit = &e; });
  return hit;
}

size_t gr_lds(const GenRowsEntry& e, int rows_lds, int kg, int halves) {
  const int cgsp = (e.nxp / 4 + 4 + 3) / 4 * 4;
  return (size_t)e.rpb * e.row_len * sizeof(float2) + (size_t)e.rpb * rows_lds * sizeof(float2) +
         (size_t)halves * kg * e.nxp * sizeof(float) + (size_t)halves * cgsp * sizeof(int) + (size_t)e.tw_lds * sizeof(float2);
}

}  // namespace

bool gen_rows_plan(int nx, int rows_lds, int kg, GenRowsPlan* plan) {
  *plan = GenRowsPlan{};
  const GenRowsEntry* best = nullptr;
  double best_cost = 1e300;
  gr_each([&](const GenRowsEntry& e) {
    if (e.r1 * e.r2 != nx) return;
    // vector instructions per row, roughly: both transforms ~ R (1 + log2 R) per lane, one wavefront pass serves RPW rows
    auto work = [](int r) { double l = 0; for (int t = r; t > 1; t >>= 1) l += 1; return r * (l + 1.0); };
    // (two rows per lane slot, GrShape::DUAL: step 1 serves all the wavefront's rows at once, step 2 runs once per row of a slot)
    const int slots = 64 / std::max(e.r1, e.r2), rw = e.rpb / GR_WAVES, ph = rw / slots;
    const double cost = (work(e.r1) + ph * work(e.r2)) / (double)rw;
    if (cost < best_cost) { best_cost = cost; best = &e; }
  });
  if (!best) return false;
  plan->r1 = best->r1;
  plan->r2 = best->r2;
  plan->rows_per_block = best->rpb;
  plan->threads = GR_THREADS;
  // the column factors are double-buffered (one barrier per candidate) unless that costs a resident workgroup
  const size_t one = gr_lds(*best, rows_lds, kg, 1), two = gr_lds(*best, rows_lds, kg, 2), cu = 160 * 1024;
  if (one > cu) return false;
  plan->halves = (two <= cu && cu / two >= std::min<size_t>(cu / one, (size_t)best->max_blocks)) ? 2 : 1;
  plan->lds = plan->halves == 2 ? two : one;
  if ((int64_t)kg * (best->nxp / 4) > (int64_t)best->pf * GR_THREADS) return false;
  return true;
}

hipError_t gen_rows_prepare(const GenRowsPlan& plan, int* blocks_per_cu) {
  const GenRowsEntry* e = gr_find(plan.r1, plan.r2);
  if (!e) return hipErrorInvalidValue;
  hipError_t rc = hipFuncSetAttribute(reinterpret_cast<const void*>(e->kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (rc != hipSuccess) return rc;
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, e->kernel, GR_THREADS, plan.lds);
}

hipError_t gen_rows_launch(const GenRowsPlan& plan, int n_ky_blocks, int layers, hipStream_t stream, const GenRowsArgs& args) {
  const GenRowsEntry* e = gr_find(plan.r1, plan.r2);
  if (!e) return hipErrorInvalidValue;
  // the kernel's indexing is compiled for this row length, this padding and this LDS layout: refuse anything else
  if (args.nx != e->r1 * e->r2 || args.nxp != e->nxp || args.kg < 1 || (int64_t)args.kg * (e->nxp / 4) > (int64_t)e->pf * GR_THREADS ||
      args.rows_lds < args.kg || plan.lds != gr_lds(*e, args.rows_lds, args.kg, plan.halves) || plan.lds > (size_t)160 * 1024 ||
      n_ky_blocks != (args.nky + e->rpb - 1) / e->rpb || layers < 1 || layers > 65535 || (args.q_out && args.q_stride < (int64_t)args.nky * args.nx))
    return hipErrorInvalidValue;
  GenRowsArgs a = args;
  a.halves = plan.halves;
  hipLaunchKernelGGL(e->kernel, dim3(n_ky_blocks, layers), dim3(GR_THREADS), plan.lds, stream, a);
  return hipGetLastError();
}
#endif  // GEN_ROWS_PART == 0
