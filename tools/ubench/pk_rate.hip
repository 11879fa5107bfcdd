// pk_rate.hip — issue rate of packed f32 VALU instructions on gfx950 (a measurement tool, not part of the library):
//   hipcc --offload-arch=gfx950 -O3 -o pk_rate pk_rate.hip && ./pk_rate
// Each kernel runs ITER rounds of 16 independent instructions per wavefront; cycles per instruction per wavefront =
// elapsed core clocks / (ITER * 16), for 1, 2 and 4 wavefronts per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
#define ITER 32768

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int KIND>
__global__ __launch_bounds__(512) void k_rate(float* out, long long* clk) {
  v2f a[16], b, c;
  b = (v2f){1.0001f + threadIdx.x * 1e-7f, 0.9999f};
  c = (v2f){1e-6f, -1e-6f};
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = (v2f){(float)i + threadIdx.x, (float)i - threadIdx.x};
  const long long t0 = clock64();
#pragma unroll 1
  for (int it = 0; it < ITER; ++it) {
    if constexpr (KIND == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
      REP16(X)
#undef X
    } else if constexpr (KIND == 1) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
      REP16(X)
#undef X
    } else if constexpr (KIND == 2) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      REP16(X)
#undef X
    } else if constexpr (KIND == 3) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      REP16(X)
#undef X
    } else if constexpr (KIND == 4) {  // swapped halves + negated high half of src1: a + (-i) c
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(a[i]) : "v"(c));
      REP16(X)
#undef X
    } else if constexpr (KIND == 5) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(c.x));
      REP16(X)
#undef X
    } else if constexpr (KIND == 6) {  // pk_fma with op_sel on two sources
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_lo:[0,0,0]" : "+v"(a[i]) : "v"(b), "v"(c));
      REP16(X)
#undef X
    } else if constexpr (KIND == 7) {  // mixed: pk_add then scalar-form fma, alternating
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %2\n\tv_fma_f32 %1, %3, %4, %1" : "+v"(a[i]), "+v"(c.y) : "v"(c), "v"(b.x), "v"(b.y));
      REP16(X)
#undef X
    }
  }
  const long long t1 = clock64();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i].x + a[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + c.y;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name, int per_instr) {
  float* out;
  long long* clk;
  hipMalloc(&out, 256 * 1024 * sizeof(float));
  hipMalloc(&clk, 1024 * sizeof(long long));
  for (int waves_per_simd : {1, 2, 4}) {
    const int threads = 64 * 4 * (waves_per_simd > 2 ? 2 : waves_per_simd > 1 ? 2 : 1);  // 256 or 512 threads
    const int blocks_per_cu = waves_per_simd == 4 ? 2 : 1;
    const int blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(threads), 0, 0, out, clk);
    hipEventRecord(e0);
    for (int w = 0; w < 4; ++w) hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(threads), 0, 0, out, clk);
    hipEventRecord(e1);
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { std::printf("launch failed\n"); return; }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 4;
    // instructions per SIMD: waves_per_simd * ITER * 16 * per_instr
    const double instr = (double)waves_per_simd * ITER * 16 * per_instr;
    std::printf("%-28s waves/SIMD %d: %.3f ms, %.2f ns per instruction per SIMD (x 2.4 GHz = %.2f cycles)\n", name, waves_per_simd,
                ms, ms * 1e6 / instr, ms * 1e6 / instr * 2.4);
  }
  hipFree(out);
  hipFree(clk);
}

int main() {
  run<0>("v_fma_f32", 1);
  run<5>("v_add_f32", 1);
  run<1>("v_pk_fma_f32", 1);
  run<2>("v_pk_add_f32", 1);
  run<3>("v_pk_mul_f32", 1);
  run<4>("v_pk_add_f32 op_sel neg_hi", 1);
  run<6>("v_pk_fma_f32 op_sel", 1);
  run<7>("v_pk_add_f32 + v_fma_f32", 2);
  return 0;
}
