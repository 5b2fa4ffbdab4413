// Stand-alone reproducer: a chain of dependent PACKED fp32 instructions with op_sel modifiers
//     v_pk_mul_f32 d, t0, w op_sel:[0,1] ; v_pk_fma_f32 d, t1, w, d op_sel_hi:[1,0,1] ; v_pk_fma_f32 d, t2, w', d op_sel:[0,1,0] ;
//     v_pk_fma_f32 d, t3, w', d op_sel_hi:[1,0,1]                      (what hipcc emits for a 4-point interpolation of float2 rows)
// returns, on MI355X, a wrong LOW half in lanes 16..31 / 48..63 -- the value short of exactly one of its four terms -- about once
// per 2e6 executions, for SOME alignments of the code and only with two or more waves per SIMD.  The same arithmetic as
// v_mul_f32 / v_fma_f32 never fails.  Each value is computed twice with each form: "the packed chain evaluated twice differs"
// counts packed-vs-packed disagreements (no reference needed), "the v_fma_f32 chain evaluated twice differs" stays 0.
//
// The kernel imitates the message pass of mol2_edge_fwd_kernel, where this was first seen (profiles/r05_mol_fused2_soak.txt):
// 512 four-wave workgroups, table rows prefetched one tile ahead (global_load_dwordx4), operands from LDS (ds_read_b128), global +
// LDS stores, a DPP maximum, a half-empty last tile, then a short MFMA phase; two rounds of that per launch.
// Variants (tools/probes/run_variants.sh, profiles/r05_pk_variants.txt): -DCHAIN=6 the chain without op_sel: clean at every padding;
// -DCHAIN=20 / 22 only the first / third instruction keeps op_sel (its LOW half reads the HIGH register of the weight pair): fail;
// -DCHAIN=21 / 23 only an op_sel_hi instruction: clean; -DNO_MFMA without the matrix phase: clean.
// -DPAD=<k> puts k dwords of s_nop ahead of the kernel body: tools/probes/run_pad_sweep.sh compiles k = 0..15 and runs each --
// on the boxes of this pool k = 0, 7, 8, 15 fail (period 32 bytes), the others are clean (profiles/r05_pk_pad_sweep.txt).
// build: hipcc --offload-arch=gfx950 -O2 -DPAD=0 tools/probes/pk_chain_probe.hip -o /tmp/pk_probe ; run: /tmp/pk_probe [launches]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#ifndef PAD
#define PAD 0
#endif
#define NF 128
#define PAD_STR2(x) #x
#define PAD_STR1(x) PAD_STR2(x)
#define PAD_STR PAD_STR1(PAD)
#define TILES 5
#define ROUNDS 2          // (message-pass-like phase, matrix phase) per launch
#define MFMA_STEPS 96     // MFMAs per wave in the matrix phase (~4 us)
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

struct Bad { unsigned long long n, low_half, lanes16_31, one_term_short, pk_twice_differ, fma_twice_differ, mix_twice_differ, mix_vs_ref; };

// SEP: 0 s_nop 0 (what the compiler emits) | 1 s_nop 1 | 2 nothing | 3 s_nop 3 | 4 s_nop 7 | 5 two s_nop 7 |
//      6 no op_sel (weights broadcast to both halves), s_nop 0 | 7 no op_sel, nothing between
#define PK_CHAIN(SEPSTR)                                                                                                          \
  asm volatile("v_pk_mul_f32 %0, %1, %5 op_sel:[0,1]\n\t" SEPSTR "v_pk_fma_f32 %0, %2, %5, %0 op_sel_hi:[1,0,1]\n\t" SEPSTR        \
               "v_pk_fma_f32 %0, %3, %6, %0 op_sel:[0,1,0]\n\t" SEPSTR "v_pk_fma_f32 %0, %4, %6, %0 op_sel_hi:[1,0,1]"             \
               : "=&v"(d) : "v"(t0), "v"(t1), "v"(t2), "v"(t3), "v"(wab), "v"(wcd))
#define PK_CHAIN_PLAIN(SEPSTR)                                                                                                    \
  asm volatile("v_pk_mul_f32 %0, %1, %5\n\t" SEPSTR "v_pk_fma_f32 %0, %2, %6, %0\n\t" SEPSTR                                       \
               "v_pk_fma_f32 %0, %3, %7, %0\n\t" SEPSTR "v_pk_fma_f32 %0, %4, %8, %0"                                             \
               : "=&v"(d) : "v"(t0), "v"(t1), "v"(t2), "v"(t3), "v"(wbb), "v"(waa), "v"(wdd), "v"(wcc))
template <int SEP>
__device__ __forceinline__ f2 chain(f2 t0, f2 t1, f2 t2, f2 t3, f2 wab, f2 wcd) {
  f2 d;
  const f2 waa = {wab.x, wab.x}, wbb = {wab.y, wab.y}, wcc = {wcd.x, wcd.x}, wdd = {wcd.y, wcd.y};
  (void)waa, (void)wbb, (void)wcc, (void)wdd;
  if (SEP == 0) PK_CHAIN("s_nop 0\n\t");
  else if (SEP == 1) PK_CHAIN("s_nop 1\n\t");
  else if (SEP == 2) PK_CHAIN("");
  else if (SEP == 3) PK_CHAIN("s_nop 3\n\t");
  else if (SEP == 4) PK_CHAIN("s_nop 7\n\t");
  else if (SEP == 5) PK_CHAIN("s_nop 7\n\ts_nop 7\n\t");
  else if (SEP == 6) PK_CHAIN_PLAIN("s_nop 0\n\t");
  else if (SEP == 20)     // only the FIRST instruction keeps its op_sel form (operands: %5 = (w.a, w.b) pair, %6.. broadcast weights)
    asm volatile("v_pk_mul_f32 %0, %1, %5 op_sel:[0,1]\n\ts_nop 0\n\tv_pk_fma_f32 %0, %2, %6, %0\n\ts_nop 0\n\tv_pk_fma_f32 %0, %3, %7, %0\n\ts_nop 0\n\tv_pk_fma_f32 %0, %4, %8, %0"
                 : "=&v"(d) : "v"(t0), "v"(t1), "v"(t2), "v"(t3), "v"(wab), "v"(waa), "v"(wdd), "v"(wcc));
  else if (SEP == 21)     // only the SECOND (op_sel_hi:[1,0,1])
    asm volatile("v_pk_mul_f32 %0, %1, %6\n\ts_nop 0\n\tv_pk_fma_f32 %0, %2, %5, %0 op_sel_hi:[1,0,1]\n\ts_nop 0\n\tv_pk_fma_f32 %0, %3, %7, %0\n\ts_nop 0\n\tv_pk_fma_f32 %0, %4, %8, %0"
                 : "=&v"(d) : "v"(t0), "v"(t1), "v"(t2), "v"(t3), "v"(wab), "v"(wbb), "v"(wdd), "v"(wcc));
  else if (SEP == 22)     // only the THIRD (op_sel:[0,1,0])
    asm volatile("v_pk_mul_f32 %0, %1, %6\n\ts_nop 0\n\tv_pk_fma_f32 %0, %2, %7, %0\n\ts_nop 0\n\tv_pk_fma_f32 %0, %3, %5, %0 op_sel:[0,1,0]\n\ts_nop 0\n\tv_pk_fma_f32 %0, %4, %8, %0"
                 : "=&v"(d) : "v"(t0), "v"(t1), "v"(t2), "v"(t3), "v"(wcd), "v"(wbb), "v"(waa), "v"(wcc));
  else if (SEP == 23)     // only the FOURTH (op_sel_hi:[1,0,1])
    asm volatile("v_pk_mul_f32 %0, %1, %6\n\ts_nop 0\n\tv_pk_fma_f32 %0, %2, %7, %0\n\ts_nop 0\n\tv_pk_fma_f32 %0, %3, %8, %0\n\ts_nop 0\n\tv_pk_fma_f32 %0, %4, %5, %0 op_sel_hi:[1,0,1]"
                 : "=&v"(d) : "v"(t0), "v"(t1), "v"(t2), "v"(t3), "v"(wcd), "v"(wbb), "v"(waa), "v"(wdd));
  else PK_CHAIN_PLAIN("");
  return d;
}

__device__ __forceinline__ float chain_ref(float t0, float t1, float t2, float t3, float wa, float wb, float wc, float wd) {
  float r;
  asm volatile("v_mul_f32 %0, %1, %6\n\ts_nop 1\n\tv_fma_f32 %0, %2, %5, %0\n\ts_nop 1\n\tv_fma_f32 %0, %3, %8, %0\n\ts_nop 1\n\tv_fma_f32 %0, %4, %7, %0"
               : "=&v"(r) : "v"(t0), "v"(t1), "v"(t2), "v"(t3), "v"(wa), "v"(wb), "v"(wc), "v"(wd));
  return r;
}

__device__ __forceinline__ float half_max(float v) {
  for (int o = 1; o < 32; o <<= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

template <int SEP>
__global__ void __launch_bounds__(256, 2) probe_kernel(const float* __restrict__ table, int rows, const float* __restrict__ nodes,
                                                       float* __restrict__ out, Bad* bad, uint32_t seed, int mfma_mode) {
#ifdef PAD
  asm volatile(".rept " PAD_STR "\n\ts_nop 0\n\t.endr" :::);
#endif
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float* sm_m = reinterpret_cast<float*>(lds);                 // 24 node rows
  float* tile = reinterpret_cast<float*>(lds + 24 * NF * 4);   // 32 rows x 132
  const int tid = threadIdx.x, lane = tid & 63, nb = tid >> 6, h = lane >> 5, c4 = 4 * (lane & 31);
  const int b = blockIdx.x;
  for (int t = tid; t < 24 * 32; t += 256) reinterpret_cast<float4*>(sm_m)[t] = reinterpret_cast<const float4*>(nodes)[(size_t)(b % 64) * 24 * 32 + t];
  __syncthreads();
  uint32_t s = seed ^ (uint32_t)(b * 2654435761u);
  auto pair_of = [&](int t, int u, int& row, float4& w, int& i, int& j) {      // per-pair "geometry": uniform over the half-wave
    uint32_t k = s ^ (uint32_t)((t * 4 + u) * 8 + 2 * nb + h) * 0x9E3779B9u;
    k ^= k >> 15, k *= 0x2c1b3c6du, k ^= k >> 12;
    row = (int)(k % (uint32_t)(rows - 4));
    const float x = (float)((k >> 8) & 1023) * (1.f / 1024.f);
    w = make_float4(-x * (x - 1.f) * (x - 2.f) * (1.f / 6.f), (x + 1.f) * (x - 1.f) * (x - 2.f) * 0.5f, -(x + 1.f) * x * (x - 2.f) * 0.5f,
                    (x + 1.f) * x * (x - 1.f) * (1.f / 6.f));
    i = (int)((k >> 3) % 24u), j = (int)((k >> 19) % 24u);
  };
  float4 T[4][4];
  auto request = [&](int t) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int row, i, j;
      float4 w;
      pair_of(t, u, row, w, i, j);
#pragma unroll
      for (int q = 0; q < 4; ++q) T[u][q] = *reinterpret_cast<const float4*>(table + (size_t)(row + q) * NF + c4);
    }
  };
  unsigned long long nbad = 0, nlow = 0, n1631 = 0, nshort = 0, npk2 = 0, nfma2 = 0, nmix2 = 0, nmixref = 0;
  float acc = 0.f;
#pragma unroll 1
  for (int round = 0; round < ROUNDS; ++round) {
  s = s * 1664525u + 1013904223u;
  request(0);
#pragma unroll 1
  for (int t = 0; t < TILES; ++t) {
    const int live = (t + 1 < TILES) ? 4 : 3;          // the last tile is partly empty, as a molecule's last tile is
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (u < live || h == 0) {
        int row, i, j;
        float4 w;
        pair_of(t, u, row, w, i, j);
        const f2 wab = {w.x, w.y}, wcd = {w.z, w.w};
        const f2 xy = chain<SEP>(f2{T[u][0].x, T[u][0].y}, f2{T[u][1].x, T[u][1].y}, f2{T[u][2].x, T[u][2].y}, f2{T[u][3].x, T[u][3].y}, wab, wcd);
        const float4 mi = *reinterpret_cast<const float4*>(sm_m + i * NF + c4);
        const f2 zw = chain<SEP>(f2{T[u][0].z, T[u][0].w}, f2{T[u][1].z, T[u][1].w}, f2{T[u][2].z, T[u][2].w}, f2{T[u][3].z, T[u][3].w}, wab, wcd);
        const float4 mj = *reinterpret_cast<const float4*>(sm_m + j * NF + c4);
        const float got[4] = {xy.x, xy.y, zw.x, zw.y};
#ifdef MIX_PROBE   // the mixed-precision fma the split-f16 kernels use: x * s - (float)(high f16 of a packed pair), op_sel on the third operand
        {
          const float ma[4] = {T[u][0].x, T[u][0].y, T[u][0].z, T[u][0].w}, mc0[4] = {T[u][1].x, T[u][1].y, T[u][1].z, T[u][1].w},
                      mc1[4] = {T[u][2].x, T[u][2].y, T[u][2].z, T[u][2].w};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            unsigned cpk;
            float r1, r2, rr, tmp;
            asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(cpk) : "v"(mc0[c]), "v"(mc1[c]));
            asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(ma[c]), "v"(w.y), "v"(cpk));
            asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r2) : "v"(ma[c]), "v"(w.y), "v"(cpk));
            asm volatile("v_lshrrev_b32 %1, 16, %4\n\tv_cvt_f32_f16 %1, %1\n\ts_nop 1\n\tv_fma_f32 %0, %2, %3, -%1" : "=&v"(rr), "=&v"(tmp) : "v"(ma[c]), "v"(w.y), "v"(cpk));
            if (__float_as_uint(r1) != __float_as_uint(r2)) ++nmix2;
            if (__float_as_uint(r1) != __float_as_uint(rr)) ++nmixref;
          }
        }
#endif
        const f2 xy2 = chain<SEP>(f2{T[u][0].x, T[u][0].y}, f2{T[u][1].x, T[u][1].y}, f2{T[u][2].x, T[u][2].y}, f2{T[u][3].x, T[u][3].y}, wab, wcd);
        const f2 zw2 = chain<SEP>(f2{T[u][0].z, T[u][0].w}, f2{T[u][1].z, T[u][1].w}, f2{T[u][2].z, T[u][2].w}, f2{T[u][3].z, T[u][3].w}, wab, wcd);
        const float got2[4] = {xy2.x, xy2.y, zw2.x, zw2.y};
        const float tq[4][4] = {{T[u][0].x, T[u][1].x, T[u][2].x, T[u][3].x}, {T[u][0].y, T[u][1].y, T[u][2].y, T[u][3].y},
                                {T[u][0].z, T[u][1].z, T[u][2].z, T[u][3].z}, {T[u][0].w, T[u][1].w, T[u][2].w, T[u][3].w}};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float want = chain_ref(tq[c][0], tq[c][1], tq[c][2], tq[c][3], w.x, w.y, w.z, w.w);
          const float want2 = chain_ref(tq[c][0], tq[c][1], tq[c][2], tq[c][3], w.x, w.y, w.z, w.w);
          if (__float_as_uint(got[c]) != __float_as_uint(got2[c])) ++npk2;
          if (__float_as_uint(want) != __float_as_uint(want2)) ++nfma2;
          if (__float_as_uint(got[c]) != __float_as_uint(want)) {
            ++nbad;
            if ((c & 1) == 0) ++nlow;
            if ((lane & 16) != 0) ++n1631;
            // short of one term?  (the four terms in the chain's order: T0 w.b, T1 w.a, T2 w.d, T3 w.c)
            const float term[4] = {tq[c][0] * w.y, tq[c][1] * w.x, tq[c][2] * w.w, tq[c][3] * w.z};
            for (int q = 0; q < 4; ++q)
              if (fabsf((want - got[c]) - term[q]) <= 1e-3f * fabsf(term[q]) + 1e-12f) { ++nshort; break; }
          }
        }
        const float4 v = make_float4(got[0] * mi.x * mj.x, got[1] * mi.y * mj.y, got[2] * mi.z * mj.z, got[3] * mi.w * mj.w);
        const int pl = 8 * u + 2 * nb + h;
        *reinterpret_cast<float4*>(out + ((size_t)b * 32 * TILES + 32 * t + pl) * NF + c4) = v;
        *reinterpret_cast<float4*>(tile + pl * 132 + c4) = v;
        acc += half_max(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
      }
    }
    __syncthreads();
    if (t + 1 < TILES) request(t + 1);
    for (int r = nb; r < 32; r += 4) acc += tile[r * 132 + lane] + tile[r * 132 + 64 + lane];       // "sums" under which the rows fly
    __syncthreads();
  }
  // the matrix phase of the edge MLPs: every wave of the workgroup on the MFMA pipe (the co-resident workgroup, dispatched a little
  // later, is still in its message pass when this starts)
#ifndef NO_MFMA
  if (mfma_mode == 0 || true) {
    f16v c0, c1;
#pragma unroll
    for (int k = 0; k < 16; ++k) c0[k] = 0.f, c1[k] = 0.f;
    h8v a, bq;
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = (_Float16)(0.001f * (float)(lane + k)), bq[k] = (_Float16)(0.002f * (float)(k + nb));
#pragma unroll 1
    for (int k = 0; k < MFMA_STEPS; k += 2) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bq, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bq, a, c1, 0, 0, 0);
    }
    acc += c0[0] + c1[3];
    __syncthreads();
  }
#endif
  }
  if (acc == 123.456f) out[0] = acc;
  if (nbad | npk2 | nfma2 | nmix2 | nmixref) {
    atomicAdd(&bad->mix_twice_differ, nmix2), atomicAdd(&bad->mix_vs_ref, nmixref);
    atomicAdd(&bad->n, nbad), atomicAdd(&bad->low_half, nlow), atomicAdd(&bad->lanes16_31, n1631), atomicAdd(&bad->one_term_short, nshort);
    atomicAdd(&bad->pk_twice_differ, npk2), atomicAdd(&bad->fma_twice_differ, nfma2);
  }
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 20000;
  const int rows = 3072, grid = 512;
  float *table, *nodes, *out;
  Bad* bad;
  CHECK(hipMalloc(&table, (size_t)rows * NF * 4));
  CHECK(hipMalloc(&nodes, (size_t)64 * 24 * NF * 4));
  CHECK(hipMalloc(&out, (size_t)grid * 32 * TILES * NF * 4));
  CHECK(hipMalloc(&bad, sizeof(Bad)));
  {
    float* h = (float*)malloc((size_t)rows * NF * 4);
    uint32_t s = 12345u;
    for (size_t i = 0; i < (size_t)rows * NF; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((float)(s >> 8) / 16777216.f - 0.5f) * expf(-(float)(i / NF) / 900.f); }
    CHECK(hipMemcpy(table, h, (size_t)rows * NF * 4, hipMemcpyHostToDevice));
    for (size_t i = 0; i < (size_t)64 * 24 * NF; ++i) { s = s * 1664525u + 1013904223u; h[i] = (float)(s >> 8) / 16777216.f - 0.5f; }
    CHECK(hipMemcpy(nodes, h, (size_t)64 * 24 * NF * 4, hipMemcpyHostToDevice));
    free(h);
  }
  const size_t lds_two = 81000, lds_one = 100000;      // two workgroups per CU / one
  typedef void (*K)(const float*, int, const float*, float*, Bad*, uint32_t, int);
#ifndef CHAIN
#define CHAIN 0      // 0: the op_sel chain with s_nop 0 between its instructions (what hipcc emits); 6: the same arithmetic without op_sel;
#endif               // 20..23: only the first / second / third / fourth instruction keeps its op_sel form
  const K kernel = probe_kernel<CHAIN>;
  CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_one));
  // (separator, matrix phase on / off, workgroups per CU)
  const int runs[][3] = {{CHAIN, 0, 2}, {CHAIN, 0, 1}};      // (chain form, .., workgroups per CU by LDS size)
  for (const auto& r : runs) {
    CHECK(hipMemset(bad, 0, sizeof(Bad)));
    const size_t lds = r[2] == 2 ? lds_two : r[2] == 1 ? lds_one : r[2] == 3 ? 52000 : 30000;
    for (int k = 0; k < launches; ++k) kernel<<<grid, 256, lds>>>(table, rows, nodes, out, bad, 77u + k, r[1]);
    CHECK(hipDeviceSynchronize());
    Bad hb;
    CHECK(hipMemcpy(&hb, bad, sizeof(Bad), hipMemcpyDeviceToHost));
    const double chains = (double)launches * grid * 256 * (TILES * 4 - 0.5) * 2 * ROUNDS;
    printf("pad %2d dwords, chain %d, LDS for %d workgroup(s) per CU: %.3g chains; packed vs v_fma_f32 chain differ %llu (%llu low half, %llu lanes 16..31 / 48..63, %llu one term short); the packed chain evaluated twice differs %llu; the v_fma_f32 chain evaluated twice differs %llu; v_fma_mix_f32 (op_sel) evaluated twice differs %llu, vs its v_fma_f32 form %llu\n",
           PAD, r[0], r[2], chains, hb.n, hb.low_half, hb.lanes16_31, hb.one_term_short, hb.pk_twice_differ, hb.fma_twice_differ, hb.mix_twice_differ, hb.mix_vs_ref);
    fflush(stdout);
  }
  return 0;
}
