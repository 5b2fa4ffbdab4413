// Probe: does `s_waitcnt vmcnt(1)` after [load (cold line), store (hot line)] guarantee that the LOAD has landed?
// The compiler assumes it does (gfx9 family: loads and stores share one in-order counter).  Each lane presets the destination
// register to a sentinel, issues a load from a random cold line, then a store to its own hot line, waits for vmcnt(1), copies the
// destination register, waits for vmcnt(0) and compares the copy with the value the buffer holds.
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/vmcnt_order_probe.hip -o gpurun_out/vmcnt_probe ; run: gpurun_out/vmcnt_probe [GiB] [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void fill_kernel(uint32_t* buf, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) buf[i] = (uint32_t)(i * 2654435761u) ^ 0x5a5a5a5au;
}

template <int WIDE>
__global__ void __launch_bounds__(256) probe_kernel(const uint32_t* cold, size_t n_words, uint32_t* hot, int iters, unsigned long long* bad,
                                                    unsigned long long* sentinel_seen, uint32_t seed) {
  const size_t gid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  uint32_t s = seed ^ (uint32_t)(gid * 747796405u + 2891336453u);
  uint32_t* my_hot = hot + gid * (WIDE ? 4 : 1);
  unsigned long long nbad = 0, nsent = 0;
  for (int it = 0; it < iters; ++it) {
    s = s * 1664525u + 1013904223u;
    // every lane its own random line (cold): 64 different lines per wave instruction
    size_t w = (((size_t)s << 7) ^ ((size_t)(s >> 3) * 0x9E3779B97F4A7C15ull)) % (n_words - 8);
    w &= ~(size_t)3;
    const uint32_t* p = cold + w;
    if (WIDE) {
      typedef uint32_t u4 __attribute__((ext_vector_type(4)));
      u4 dst = {0xDEADBEEFu, 0xDEADBEEFu, 0xDEADBEEFu, 0xDEADBEEFu};
      u4 val = {s, s, s, s};
      __shared__ u4 caps[256];
      const uint32_t lds_addr = (uint32_t)(uintptr_t)(caps + threadIdx.x);     // (ds_write reads its source registers at issue)
      asm volatile(
          "global_load_dwordx4 %0, %1, off\n\t"
          "global_store_dwordx4 %2, %3, off\n\t"
          "s_waitcnt vmcnt(1)\n\t"
          "ds_write_b128 %4, %0\n\t"
          "s_waitcnt vmcnt(0) lgkmcnt(0)"
          : "+v"(dst)
          : "v"(p), "v"(my_hot), "v"(val), "v"(lds_addr)
          : "memory");
      const u4 cap = caps[threadIdx.x];
      for (int c = 0; c < 4; ++c) {
        const uint32_t want = (uint32_t)((w + c) * 2654435761u) ^ 0x5a5a5a5au;
        const uint32_t got = c == 0 ? cap.x : c == 1 ? cap.y : c == 2 ? cap.z : cap.w;
        if (got != want) { ++nbad; if (got == 0xDEADBEEFu) ++nsent; }
      }
    } else {
      uint32_t dst = 0xDEADBEEFu, cap;
      asm volatile(
          "global_load_dword %0, %2, off\n\t"
          "global_store_dword %3, %4, off\n\t"
          "s_waitcnt vmcnt(1)\n\t"
          "v_mov_b32 %1, %0\n\t"
          "s_waitcnt vmcnt(0)"
          : "+v"(dst), "=&v"(cap)
          : "v"(p), "v"(my_hot), "v"(s)
          : "memory");
      const uint32_t want = (uint32_t)(w * 2654435761u) ^ 0x5a5a5a5au;
      if (cap != want) { ++nbad; if (cap == 0xDEADBEEFu) ++nsent; }
    }
  }
  if (nbad) atomicAdd(bad, nbad);
  if (nsent) atomicAdd(sentinel_seen, nsent);
}

int main(int argc, char** argv) {
  const double gib = argc > 1 ? atof(argv[1]) : 8.0;
  const int iters = argc > 2 ? atoi(argv[2]) : 2000;
  const size_t n_words = (size_t)(gib * (1ull << 30)) / 4;
  uint32_t *cold, *hot;
  unsigned long long *cnt, h[2];
  const int blocks = 2048, threads = 256;
  CHECK(hipMalloc(&cold, n_words * 4));
  CHECK(hipMalloc(&hot, (size_t)blocks * threads * 16));
  CHECK(hipMalloc(&cnt, 16));
  fill_kernel<<<4096, 256>>>(cold, n_words);
  CHECK(hipDeviceSynchronize());
  for (int wide = 0; wide < 2; ++wide) {
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipMemset(cnt, 0, 16));
      if (wide) probe_kernel<1><<<blocks, threads>>>(cold, n_words, hot, iters, cnt, cnt + 1, 1234u + rep);
      else probe_kernel<0><<<blocks, threads>>>(cold, n_words, hot, iters, cnt, cnt + 1, 1234u + rep);
      CHECK(hipDeviceSynchronize());
      CHECK(hipMemcpy(h, cnt, 16, hipMemcpyDeviceToHost));
      printf("%s loads, rep %d: %llu lane-values checked after vmcnt(1), %llu wrong (%llu still the sentinel)\n", wide ? "dwordx4" : "dword", rep,
             (unsigned long long)blocks * threads * iters * (wide ? 4 : 1), h[0], h[1]);
    }
  }
  return 0;
}
