// stale_probe.hip — does a small, hot word survive a launch boundary in a cache of the compute unit?
//
// Round 3 saw the batched Path-A solver's per-candidate state come back STALE across launches on one stream: a scalar
// step read-modify-wrote a few words every LSMR iteration and, about once in a thousand candidate-solves, read the value
// it had written two launches earlier.  Agent-scope fences inside the kernels did not cure it; agent-scope atomic loads
// did.  This probe isolates the access pattern: CELLS words, one 64-lane workgroup per word and launch, every launch
// increments its word.  Two kernels alternate (like the solver's two scalar steps), a filler kernel keeps the other
// compute units busy.  After LAUNCHES launches every word must equal LAUNCHES; a smaller value is a lost update, i.e. a
// stale read.  Variants of the READ:
//   0  plain load through a const __restrict__ pointer at a workgroup-uniform address (the compiler may use s_load:
//      scalar data cache)
//   1  plain load at a per-lane address (vector L1 / L2)
//   2  plain load as in 0 behind __builtin_amdgcn_fence(acquire, "agent")
//   3  agent-scope atomic load (what the solver uses now)
// build: hipcc --offload-arch=gfx950 -O3 -o stale_probe stale_probe.hip ; run: ./stale_probe [cells] [launches]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

struct alignas(128) Cell {
  long long value;
  long long pad[15];
};

template <int VARIANT, int WHICH>
__global__ __launch_bounds__(64) void k_step(Cell* cells, const Cell* __restrict__ ro, int n) {
  const int c = blockIdx.x;
  if (c >= n) return;
  long long v;
  if (VARIANT == 0) {
    v = ro[c].value;
  } else if (VARIANT == 1) {
    const long long* q = &cells[c].value;   // a plain vector load (no cache-policy bits), whatever the compiler knows about uniformity
    asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(q) : "memory");
  } else if (VARIANT == 2) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    v = ro[c].value;
  } else {
    v = __hip_atomic_load(&cells[c].value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // a little dependent arithmetic, like the solver's recurrences
  long long w = v;
#pragma unroll 1
  for (int k = 0; k < 8 + WHICH; ++k) w = (w * 3 + 1) / 3;
  if (threadIdx.x == 0) cells[c].value = v + 1 + (w - w);
}

__global__ __launch_bounds__(256) void k_fill(float* buf, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) buf[i] = buf[i] * 1.0001f + 1.f;
}

template <int VARIANT>
long long run(int cells, int launches, float* fill, size_t nfill, hipStream_t s) {
  Cell* d = nullptr;
  if (hipMalloc(&d, sizeof(Cell) * cells) != hipSuccess) return -1;
  (void)hipMemsetAsync(d, 0, sizeof(Cell) * cells, s);
  for (int t = 0; t < launches; t += 2) {
    hipLaunchKernelGGL((k_step<VARIANT, 0>), dim3(cells), dim3(64), 0, s, d, d, cells);
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, s, fill, nfill);
    hipLaunchKernelGGL((k_step<VARIANT, 1>), dim3(cells), dim3(64), 0, s, d, d, cells);
  }
  std::vector<Cell> h(cells);
  (void)hipMemcpyAsync(h.data(), d, sizeof(Cell) * cells, hipMemcpyDeviceToHost, s);
  (void)hipStreamSynchronize(s);
  (void)hipFree(d);
  long long lost = 0;
  for (int c = 0; c < cells; ++c) lost += launches - h[c].value;
  return lost;
}

int main(int argc, char** argv) {
  const int cells = argc > 1 ? std::atoi(argv[1]) : 256;
  const int launches = argc > 2 ? std::atoi(argv[2]) & ~1 : 4000;
  hipStream_t s;
  if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { std::printf("no device\n"); return 1; }
  const size_t nfill = (size_t)8 << 20;
  float* fill = nullptr;
  (void)hipMalloc(&fill, nfill * sizeof(float));
  (void)hipMemsetAsync(fill, 0, nfill * sizeof(float), s);
  std::printf("cells %d launches %d (updates expected per variant: %lld)\n", cells, launches, (long long)cells * launches);
  std::printf("variant 0 (uniform plain load)          lost updates: %lld\n", run<0>(cells, launches, fill, nfill, s));
  std::printf("variant 1 (per-lane plain load)         lost updates: %lld\n", run<1>(cells, launches, fill, nfill, s));
  std::printf("variant 2 (acquire fence + plain load)  lost updates: %lld\n", run<2>(cells, launches, fill, nfill, s));
  std::printf("variant 3 (agent-scope atomic load)     lost updates: %lld\n", run<3>(cells, launches, fill, nfill, s));
  (void)hipFree(fill);
  return 0;
}
