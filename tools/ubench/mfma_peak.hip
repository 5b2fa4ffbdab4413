// Microbenchmark: sustained v_mfma_f32_32x32x2_f32 rate and shader clock on this box (random operands).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int LDSB>
__global__ void __launch_bounds__(256) k(const float* in, float* out, int iters, long long* clk) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = in[i];
  __syncthreads();
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = in[threadIdx.x * 8 + i]; b[i] = in[2048 + threadIdx.x * 8 + i]; }
  f32x16 acc[NACC];
  for (int n = 0; n < NACC; ++n) for (int j = 0; j < 16; ++j) acc[n][j] = 0.f;
  long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float bb = b[i];
      if (LDSB) bb = lds[(threadIdx.x * 4 + i * 64 + it) & 4095];
#pragma unroll
      for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bb, acc[n], 0, 0, 0);
    }
  }
  long long c1 = clock64(), w1 = wall_clock64();
  float s = 0;
  for (int n = 0; n < NACC; ++n) for (int j = 0; j < 16; ++j) s += acc[n][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

template <int NACC, int LDSB>
void run(int blocks, const char* name, const float* din, float* dout, long long* dclk) {
  int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC, LDSB><<<blocks, 256>>>(din, dout, 100, dclk);
  hipEventRecord(e0);
  k<NACC, LDSB><<<blocks, 256>>>(din, dout, iters, dclk);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long clk[2]; hipMemcpy(clk, dclk, 16, hipMemcpyDeviceToHost);
  double n_mfma = (double)blocks * 4 * iters * 8 * NACC;
  double tf = n_mfma * 4096 / (ms * 1e-3) / 1e12;
  double ghz = (double)clk[0] / ((double)clk[1] / 100e6) / 1e9;
  printf("%-28s blocks %4d  %.3f ms  %.1f TF  clock64/wall = %.3f GHz  cycles/MFMA/SIMD = %.1f\n", name, blocks, ms, tf, ghz,
         (double)clk[0] / (iters * 8.0 * NACC) / (blocks > 256 ? blocks / 256.0 : 1.0));
}

int main() {
  float* h = (float*)malloc(4096 * 4);
  for (int i = 0; i < 4096; ++i) h[i] = (float)rand() / RAND_MAX * 2 - 1;
  float *din, *dout; long long* dclk;
  hipMalloc(&din, 4096 * 4); hipMalloc(&dout, 1024 * 256 * 4); hipMalloc(&dclk, 16);
  hipMemcpy(din, h, 4096 * 4, hipMemcpyHostToDevice);
  run<4, 0>(256, "4 acc, 1 wave/SIMD", din, dout, dclk);
  run<4, 0>(512, "4 acc, 2 waves/SIMD", din, dout, dclk);
  run<1, 0>(256, "1 acc (dependent), 1 w/SIMD", din, dout, dclk);
  run<1, 0>(512, "1 acc (dependent), 2 w/SIMD", din, dout, dclk);
  run<1, 1>(512, "1 acc + LDS B, 2 w/SIMD", din, dout, dclk);
  run<4, 1>(256, "4 acc + LDS B, 1 w/SIMD", din, dout, dclk);
  memset(h, 0, 4096 * 4); hipMemcpy(din, h, 4096 * 4, hipMemcpyHostToDevice);
  run<4, 0>(256, "ZERO data 4 acc 1 w/SIMD", din, dout, dclk);
  return 0;
}
