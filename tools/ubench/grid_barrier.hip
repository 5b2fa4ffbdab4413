// Micro-benchmark: cost of a software grid barrier (agent-scope atomics) on gfx950 against the launch-to-launch latency of
// dependent empty kernels.  Variants: participating workgroups spread over all XCDs, or only the blocks with b % 8 == 0
// (round-robin dispatch puts those on one XCD).  Spin loops are bounded: a lost barrier sets a flag instead of hanging the GPU.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/grid_barrier.hip -o tools/ubench/grid_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ bool grid_sync(unsigned* counter, unsigned target, int* fail) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE);   // agent scope by default in HIP for __atomic on global
    unsigned spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > (1u << 22)) { *fail = 1; ok = false; break; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  return ok;
}

// stride 8: only blocks with blockIdx.x % 8 == 0 take part (one XCD); stride 1: all blocks
__global__ void __launch_bounds__(512) barrier_kernel(unsigned* counter, int n_iter, int stride, int members, float* data, int* fail) {
  if (blockIdx.x % stride) return;
  const int me = blockIdx.x / stride;
  float v = 0.f;
  for (int it = 0; it < n_iter; ++it) {
    // a token of dependent work: every member writes a value its right neighbour reads after the barrier
    if (threadIdx.x == 0) __hip_atomic_store(data + me, (float)it + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!grid_sync(counter, (unsigned)members * (it + 1), fail)) return;
    v = __hip_atomic_load(data + (me + 1) % members, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * 1e-9f;
  }
  if (threadIdx.x == 0) data[members + me] = v;
}
__global__ void empty_kernel(float* data) { if (threadIdx.x == 1000) data[0] = 1.f; }

int main() {
  unsigned* counter;
  float* data;
  int* fail;
  CHECK(hipMalloc(&counter, 4));
  CHECK(hipMalloc(&data, 4096 * 4));
  CHECK(hipMalloc(&fail, 4));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  const int n_iter = 2000;
  for (int stride = 1; stride <= 8; stride *= 8)
    for (int members = 4; members <= 32; members *= 2) {
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemset(counter, 0, 4));
        CHECK(hipMemset(fail, 0, 4));
        CHECK(hipEventRecord(a));
        barrier_kernel<<<members * stride, 512>>>(counter, n_iter, stride, members, data, fail);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
      }
      int f;
      CHECK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
      printf("stride %d members %2d: %.2f us per barrier%s\n", stride, members, best * 1e3f / n_iter, f ? "  (BARRIER LOST)" : "");
    }
  CHECK(hipEventRecord(a));
  for (int i = 0; i < 2000; ++i) empty_kernel<<<1, 64>>>(data);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms;
  CHECK(hipEventElapsedTime(&ms, a, b));
  printf("dependent empty kernels: %.2f us per launch\n", ms * 1e3f / 2000);
  return 0;
}
