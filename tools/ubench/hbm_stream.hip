// Microbenchmark: achievable HBM read / write / copy bandwidth on this box with plain float4 streaming kernels
// (the practical ceiling the edge kernels' L2-miss traffic can be compared with).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/hbm_stream.hip -o /tmp/hbm_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void __launch_bounds__(256) k_read(const float4* __restrict__ a, size_t n, float* out) {
  float4 s = make_float4(0, 0, 0, 0);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float4 v = a[i];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  if (s.x + s.y + s.z + s.w == 12345.f) out[0] = s.x;
}
__global__ void __launch_bounds__(256) k_write(float4* __restrict__ a, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a[i] = make_float4(1, 2, 3, 4);
}
__global__ void __launch_bounds__(256) k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
int main() {
  for (size_t mb : {80, 320, 1280}) {
    const size_t n = mb * 1024 * 1024 / 16;
    float4 *a, *b; float* out;
    hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMalloc(&out, 4);
    hipMemset(a, 0, n * 16); hipMemset(b, 0, n * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {2048, 8192}) {
      float ms;
      for (int w = 0; w < 2; ++w) k_read<<<blocks, 256>>>(a, n, out);
      hipEventRecord(e0); for (int r = 0; r < 10; ++r) k_read<<<blocks, 256>>>(a, n, out); hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1); const double rd = 10.0 * n * 16 / (ms * 1e-3) / 1e12;
      hipEventRecord(e0); for (int r = 0; r < 10; ++r) k_write<<<blocks, 256>>>(b, n); hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1); const double wr = 10.0 * n * 16 / (ms * 1e-3) / 1e12;
      hipEventRecord(e0); for (int r = 0; r < 10; ++r) k_copy<<<blocks, 256>>>(a, b, n); hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1); const double cp = 20.0 * n * 16 / (ms * 1e-3) / 1e12;
      printf("%5zu MB  %5d blocks: read %.2f TB/s  write %.2f TB/s  copy (r+w) %.2f TB/s\n", mb, blocks, rd, wr, cp);
    }
    hipFree(a); hipFree(b); hipFree(out);
  }
  return 0;
}
