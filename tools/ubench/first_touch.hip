// Micro-benchmark: what a FIRST touch costs inside a small kernel on gfx950.  One wave issues K independent 16-byte-per-lane
// loads whose addresses are `stride` bytes apart (so K distinct lines / pages / 2 MB fragments), then waits for all of them;
// wall-clock (100 MHz) stamps give the issue time and the completion time.  Every launch uses a fresh offset so that nothing is
// warm from the launch before; a second pass inside the same kernel repeats the same addresses (warm TLB, warm L2).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/first_touch.hip -o tools/ubench/first_touch
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int K>
__global__ void __launch_bounds__(64) touch_kernel(const float* base, size_t stride_f, size_t offset_f, long long* out, float* sink) {
  const float* p = base + offset_f + 4 * threadIdx.x;
  float4 v[K];
  long long t[6];
  t[0] = wall_clock64();
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = *reinterpret_cast<const float4*>(p + k * stride_f);
  __builtin_amdgcn_sched_barrier(0);
  t[1] = wall_clock64();
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) s += v[k].x + v[k].w;
  __builtin_amdgcn_sched_barrier(0);
  t[2] = wall_clock64();
  // warm repeat: same pages, the neighbouring 1 KB
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = *reinterpret_cast<const float4*>(p + k * stride_f + 256);
  __builtin_amdgcn_sched_barrier(0);
  t[3] = wall_clock64();
#pragma unroll
  for (int k = 0; k < K; ++k) s += v[k].y + v[k].z;
  __builtin_amdgcn_sched_barrier(0);
  t[4] = wall_clock64();
  if (threadIdx.x == 0) {
    for (int k = 0; k < 5; ++k) out[k] = t[k];
  }
  if (s == 12345.678f) sink[threadIdx.x] = s;
}

int main() {
  const size_t bytes = (size_t)1 << 30;
  float* buf;
  long long* out;
  float* sink;
  CHECK(hipMalloc(&buf, bytes));
  CHECK(hipMemset(buf, 0, bytes));
  CHECK(hipMalloc(&out, 64));
  CHECK(hipMalloc(&sink, 1024));
  const size_t strides[] = {1024, 4096, 65536, (size_t)2 << 20, (size_t)32 << 20};
  for (size_t st : strides) {
    double acc[4] = {0, 0, 0, 0};
    const int reps = 20;
    for (int r = 0; r < reps; ++r) {
      // a fresh window every launch: nothing of it was touched by the previous launches
      const size_t off = ((size_t)r * 16 * st + (size_t)r * 8192) % (bytes / 2);
      if (off + 16 * st + 4096 > bytes) continue;
      touch_kernel<16><<<1, 64>>>(buf, st / 4, off / 4, out, sink);
      long long h[5];
      CHECK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
      for (int k = 0; k < 4; ++k) acc[k] += (h[k + 1] - h[k]) / 100.0;
    }
    printf("stride %9zu B, 16 loads: cold issue %.2f us, cold wait %.2f us | warm issue %.2f us, warm wait %.2f us\n", st,
           acc[0] / reps, acc[1] / reps, acc[2] / reps, acc[3] / reps);
  }
  return 0;
}
