// Microbenchmark: how many cycles does one SIMD of gfx950 need per wave64 v_fma_f32, as a function of the
// waves resident per SIMD?  (Settles how to read SQ_ACTIVE_INST_VALU: is a kernel whose summed per-wave VALU
// activity equals one quad-cycle per SIMD cycle at the issue limit, or at half of it?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_fma(float* out, long long* cyc, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float b = 1.0001f, c = 0.5f;
  __syncthreads();
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a0 = fmaf(a0, b, c); a1 = fmaf(a1, b, c); a2 = fmaf(a2, b, c); a3 = fmaf(a3, b, c);
      a4 = fmaf(a4, b, c); a5 = fmaf(a5, b, c); a6 = fmaf(a6, b, c); a7 = fmaf(a7, b, c);
    }
  }
  const long long t1 = clock64();
  if (threadIdx.x % 64 == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount, iters = 4096;
  float* out; long long* cyc;
  hipMalloc(&out, (size_t)cus * 32 * 64 * sizeof(float));
  hipMalloc(&cyc, (size_t)cus * 32 * sizeof(long long));
  for (int wps : {1, 2, 3, 4, 6, 8}) {           // waves per SIMD: blocks of 256 threads = 1 wave per SIMD each
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_fma<<<cus * wps, 256>>>(out, cyc, 16);     // warm
    hipEventRecord(e0);
    k_fma<<<cus * wps, 256>>>(out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h((size_t)cus * wps * 4);
    hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += v; mean /= h.size();
    const double instr = (double)iters * 64;     // v_fma per wave
    // clock64 ticks at a fixed 100 MHz on this part; report both the tick-based and the wall-based figures
    printf("waves/SIMD %d: kernel %.3f ms, %.2f ns per wave-instruction per SIMD (wall), clock64 ticks per wave-instr %.4f\n",
           wps, ms, ms * 1e6 / (instr * wps), mean / instr);
  }
  return 0;
}
