// Row-local fused node kernels (node128.hip: node_fwd_kernel / node_bwd_kernel) with split-f16 products on the matrix cores.
//
// Same stages, same tiling (one 4-wave workgroup per 32 atom rows, wave w = output block w of every GEMM, hand-over through an
// LDS tile) and the same results contract as the fp32 kernels; the products take the form of mlp128s.hip: every fp32 operand
// as two scaled f16 pieces, hi_a hi_b + hi_a lo_b + lo_a hi_b on v_mfma_f32_32x32x16_f16 with fp32 accumulation (5.3x less
// matrix-pipe time than the v_mfma_f32_32x32x2_f32 chain, and closer to fp64).
//   weights : f16 (hi, lo) images prepared once per parameter set (weight_image_kernel; pipeline.hip keeps them in the
//             prepared block), scaled per matrix, stored in MFMA fragment order -- a wave's A fragments of one GEMM are 16
//             16-byte loads per lane, each a contiguous 1 KiB of the image;
//   rows    : scaled per row by the largest magnitude of the row.  A row's 128 values live in 8 lanes of 4 waves: each lane
//             publishes the maximum of its 16 values in LDS BEFORE the barrier that already separates two uses of the tile, so the
//             exchange costs no extra barrier; the (hi, lo) pieces are written to the LDS tile in fragment order.
#include <stdint.h>
#include <string.h>

#include "common.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef __bf16 b4 __attribute__((ext_vector_type(4)));

#ifndef NS_WG_PER_CU
#define NS_WG_PER_CU 2   // launch-bounds hint: workgroups per CU (tooling: 3 = every tile of config 2 resident at once, 168 VGPRs)
#endif
#define NS_PITCH 272                         // bytes per row of one plane of the LDS tile (128 f16 + 16 pad)
#define NS_PLANE (32 * NS_PITCH)
#define NS_LDS_BYTES (2 * NS_PLANE + 8 * 32 * 4)
#ifndef NS_MERGE3
#define NS_MERGE3 0   // 1 (tooling): the three components of equiv_update (and of its adjoint) share ONE publish / commit / barrier
                      // round -- three LDS tiles, three GEMMs behind the same weight fragments.  Measured 3 % SLOWER (0.308 vs
                      // 0.298 ms per step for the six launches): the kernels are bound by their HBM streams, not the stage count
#endif
#define WIMG_PLANE (NF * NF * 2)             // one f16 plane of a weight image

// ---- weight images ----------------------------------------------------------------------------------------------
// src: fp32 [128][128] row-major (rows = output features of D^T = W . X^T).  dst: hi plane, lo plane, then the inverse scale as one
// float.  A plane is stored in FRAGMENT order: 32 blocks (output block nb, MFMA step T) of 1 KiB, lane 32 h + r of a block
// holding the 8 f16 that lane feeds to MFMA T as the A operand of output row 32 nb + r (k-slots 16 T + 8 h .. + 7; slot
// 16 T + 8 h + 4 j + c <-> input feature 16 T + 8 j + 4 h + c).  One fragment load of a wave is then 1 KiB contiguous -- 8 cache
// lines; with the rows of a [out][k-slot] matrix it was 32 lines of which 32 bytes each were used, and the node kernels spent
// ~1.1 us of address-unit time per matrix on it (5 matrices per launch; 8x the requests of the activations they stream).
__device__ __forceinline__ void ns_pow2_scale(float m, float& S, float& inv) {
  const int e = (int)((__float_as_uint(m) >> 23) & 0xffu);
  const bool ok = e >= 40 && e < 255;
  S = ok ? __uint_as_float((unsigned)(268 - e) << 23) : 1.0f;
  inv = ok ? __uint_as_float((unsigned)(e - 14) << 23) : 1.0f;
}
__global__ void __launch_bounds__(1024) weight_image_kernel(WeightImageJobs jobs) {
  __shared__ float red[16];
  const float* src = jobs.src[blockIdx.x];
  char* dst = jobs.dst[blockIdx.x];
  float4 v[4];
  float m = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    v[q] = reinterpret_cast<const float4*>(src)[threadIdx.x + 1024 * q];
    m = fmaxf(fmaxf(fmaxf(m, fabsf(v[q].x)), fmaxf(fabsf(v[q].y), fabsf(v[q].z))), fabsf(v[q].w));
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
#pragma unroll
  for (int w = 0; w < 16; ++w) m = fmaxf(m, red[w]);
  float S, inv;
  ns_pow2_scale(m, S, inv);
  const bool bf = jobs.fmt == WIMG_FMT_BF16;     // (uniform) bf16 has fp32's exponent range: no scale
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int idx = threadIdx.x + 1024 * q, o = idx >> 5, c = idx & 31;
    // fragment (nb = o >> 5, T = c >> 2), lane 32 h + r (h = c & 1, r = o & 31), second half of the lane's 16 bytes when c & 2
    const int off = ((((o >> 5) * 8 + (c >> 2)) * 64 + (c & 1) * 32 + (o & 31)) << 4) + ((c >> 1) & 1) * 8;
    if (bf) {
      b4 w;
      w[0] = (__bf16)v[q].x, w[1] = (__bf16)v[q].y, w[2] = (__bf16)v[q].z, w[3] = (__bf16)v[q].w;
      *reinterpret_cast<b4*>(dst + off) = w;
      continue;
    }
    const float s[4] = {v[q].x * S, v[q].y * S, v[q].z * S, v[q].w * S};
    h4 hi, lo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const _Float16 a = (_Float16)s[j];
      hi[j] = a;
      lo[j] = (_Float16)(s[j] - (float)a);
    }
    *reinterpret_cast<h4*>(dst + off) = hi;
    *reinterpret_cast<h4*>(dst + WIMG_PLANE + off) = lo;
  }
  if (threadIdx.x == 0) {
    *reinterpret_cast<float*>(dst + 2 * WIMG_PLANE) = bf ? 1.0f : inv;
    *reinterpret_cast<int*>(dst + 2 * WIMG_PLANE + 4) = jobs.fmt;
  }
}
int launch_weight_images(const float* const* src, char* const* dst, int n, hipStream_t s, int fmt) {
  for (int o = 0; o < n; o += WIMG_MAX_JOBS) {
    WeightImageJobs jobs;
    jobs.fmt = fmt;
    const int c = n - o < WIMG_MAX_JOBS ? n - o : WIMG_MAX_JOBS;
    for (int k = 0; k < c; ++k) {
      jobs.src[k] = src[o + k];
      jobs.dst[k] = dst[o + k];
    }
    weight_image_kernel<<<c, 1024, 0, s>>>(jobs);
    LAUNCH_CHECK();
  }
  return 0;
}

// ---- tile helpers -----------------------------------------------------------------------------------------------
struct STile {
  char* img;     // LDS: hi plane, lo plane of the 32-row tile
  float* pmax;   // LDS [8][32]: per (wave, lane half) maxima of each row
  int r, h, nb;
};
struct WFrag {
  h8 hi[8], lo[8];
  float inv;     // inverse scale of the matrix
};

// A fragments of output block nb (rows nb*32 + r of the image), fetched one GEMM ahead like node128.hip:load_w
__device__ __forceinline__ void load_wimg(WFrag& w, const STile& t, const char* __restrict__ img) {
  const char* p = img + ((size_t)(t.nb * 8 * 64 + t.h * 32 + t.r) << 4);
#pragma unroll
  for (int T = 0; T < 8; ++T) {
#ifdef NS_ABL_HOT_W   // tooling (wrong results): every fragment from the first KiB of the image -- the price of streaming the weights from L2
    w.hi[T] = *reinterpret_cast<const h8*>(img + ((t.h * 32 + t.r) << 4));
    w.lo[T] = *reinterpret_cast<const h8*>(img + ((t.h * 32 + t.r) << 4));
#else
    w.hi[T] = *reinterpret_cast<const h8*>(p + 1024 * T);
    w.lo[T] = *reinterpret_cast<const h8*>(p + WIMG_PLANE + 1024 * T);
#endif
  }
  w.inv = *reinterpret_cast<const float*>(img + 2 * WIMG_PLANE);
  __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ float row_amax16(const float (&v)[16]) {
  float m = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) m = fmaxf(m, fabsf(v[k]));
  return m;
}
// publish this lane's share of the row maximum (before the barrier that frees the tile)
__device__ __forceinline__ void tile_publish(const float (&v)[16], const STile& t) { t.pmax[(t.nb * 2 + t.h) * 32 + t.r] = row_amax16(v); }
// after that barrier: row scale, split, write the fragment-ordered planes; returns the inverse row scale.
// v[4 q + c] = feature nb*32 + 8 q + 4 h + c  ->  k-slot (2 nb + (q >> 1)) * 16 + 8 h + 4 (q & 1) + c
__device__ __forceinline__ float tile_commit(const float (&v)[16], const STile& t) {
  float m = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) m = fmaxf(m, t.pmax[j * 32 + t.r]);
  float S, inv;
  ns_pow2_scale(m, S, inv);
  char* row = t.img + t.r * NS_PITCH + 2 * (t.nb * 32 + 8 * t.h);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    h4 hi, lo;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float s = v[4 * q + c] * S;
      const _Float16 a = (_Float16)s;
      hi[c] = a;
      lo[c] = (_Float16)(s - (float)a);
    }
    const int off = 2 * ((q >> 1) * 16 + (q & 1) * 4);
    *reinterpret_cast<h4*>(row + off) = hi;
    *reinterpret_cast<h4*>(row + NS_PLANE + off) = lo;
  }
  return inv;
}

// block nb of D^T = W . X^T for the 32 rows of the LDS tile, scaled back: B fragments two triples ahead of their use
__device__ __forceinline__ void tile_gemm_s(float (&out)[16], const STile& t, const WFrag& w, float inv_row) {
  f32x16 acc;
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.f;
  const char* xr = t.img + t.r * NS_PITCH + 16 * t.h;
  h8 bh0 = *reinterpret_cast<const h8*>(xr), bl0 = *reinterpret_cast<const h8*>(xr + NS_PLANE);
  h8 bh1 = *reinterpret_cast<const h8*>(xr + 32), bl1 = *reinterpret_cast<const h8*>(xr + NS_PLANE + 32);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int T = 0; T < 8; ++T) {
    h8 bh2, bl2;
    if (T < 6) {
      bh2 = *reinterpret_cast<const h8*>(xr + 32 * (T + 2));
      bl2 = *reinterpret_cast<const h8*>(xr + NS_PLANE + 32 * (T + 2));
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.hi[T], bh0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.hi[T], bl0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.lo[T], bh0, acc, 0, 0, 0);
    bh0 = bh1;
    bl0 = bl1;
    if (T < 6) {
      bh1 = bh2;
      bl1 = bl2;
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
  const float sc = inv_row * w.inv;
#pragma unroll
  for (int k = 0; k < 16; ++k) out[k] = acc[k] * sc;
}

// ---- bf16 operands (training under autocast(bfloat16)): the tile holds ONE plane of bf16 values in the same fragment order, no
// row scale (bf16 has fp32's exponent range), and a GEMM step is one v_mfma_f32_32x32x16_bf16 instead of three f16 ones
__device__ __forceinline__ void tile_commit_bf(const float (&v)[16], const STile& t) {
  char* row = t.img + t.r * NS_PITCH + 2 * (t.nb * 32 + 8 * t.h);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    b4 w;
#pragma unroll
    for (int c = 0; c < 4; ++c) w[c] = (__bf16)v[4 * q + c];
    *reinterpret_cast<b4*>(row + 2 * ((q >> 1) * 16 + (q & 1) * 4)) = w;
  }
}
__device__ __forceinline__ void tile_gemm_bf(float (&out)[16], const STile& t, const WFrag& w) {
  f32x16 acc;
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.f;
  const char* xr = t.img + t.r * NS_PITCH + 16 * t.h;
  b8 b0 = *reinterpret_cast<const b8*>(xr), b1 = *reinterpret_cast<const b8*>(xr + 32);
#pragma unroll
  for (int T = 0; T < 8; ++T) {
    b8 b2;
    if (T < 6) b2 = *reinterpret_cast<const b8*>(xr + 32 * (T + 2));
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, w.hi[T]), b0, acc, 0, 0, 0);
    b0 = b1;
    if (T < 6) b1 = b2;
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) out[k] = acc[k];
}
// the format word of an image (uniform: a scalar load)
__device__ __forceinline__ bool wimg_is_bf16(const char* __restrict__ img) {
  return *reinterpret_cast<const int*>(img + 2 * WIMG_PLANE + 4) == WIMG_FMT_BF16;
}

// this lane's 16 values of column block nb (features nb*32 + (k&3) + 8 (k>>2) + 4h) of one row
__device__ __forceinline__ void sblk_load(float (&v)[16], const float* __restrict__ base, size_t row_off, const STile& t) {
  const float4* p = reinterpret_cast<const float4*>(base + row_off + t.nb * 32 + 4 * t.h);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 x = p[2 * q];
    v[4 * q] = x.x;
    v[4 * q + 1] = x.y;
    v[4 * q + 2] = x.z;
    v[4 * q + 3] = x.w;
  }
}
__device__ __forceinline__ void sblk_store(const float (&v)[16], float* __restrict__ base, size_t row_off, const STile& t) {
  float4* p = reinterpret_cast<float4*>(base + row_off + t.nb * 32 + 4 * t.h);
#pragma unroll
  for (int q = 0; q < 4; ++q) p[2 * q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}

#ifdef NS_CLOCK_DEBUG   // tooling: wall-clock (100 MHz) and shader-clock stamps around every GEMM of one workgroup
#define NS_DBG_INIT() long long dbg_w[16], dbg_c[16]; int dbg_n = 0; dbg_w[0] = wall_clock64(); dbg_c[0] = clock64(); dbg_n = 1;
#define NS_DBG_STAMP() if (dbg_n < 16) { dbg_w[dbg_n] = wall_clock64(); dbg_c[dbg_n] = clock64(); ++dbg_n; }
#define NS_DBG_PRINT(tag)                                                                                      \
  NS_DBG_STAMP()                                                                                               \
  if (blockIdx.x == 0 && threadIdx.x == 0) {                                                                   \
    printf(tag " start %lld:", dbg_w[0]);                                                                      \
    for (int k_ = 1; k_ < dbg_n; ++k_)                                                                         \
      printf(" +%.2fus(%.2fGHz)", (dbg_w[k_] - dbg_w[k_ - 1]) / 100.0,                                         \
             (dbg_c[k_] - dbg_c[k_ - 1]) / ((dbg_w[k_] - dbg_w[k_ - 1]) * 10.0 + 1e-9));                        \
    printf("  end %lld\n", dbg_w[dbg_n - 1]);                                                                  \
  }
#else
#define NS_DBG_INIT()
#define NS_DBG_STAMP()
#define NS_DBG_PRINT(tag)
#endif
#define NS_TILE_SETUP()                                                      \
  __shared__ __attribute__((aligned(16))) char lds[(NS_MERGE3 ? 3 : 1) * NS_LDS_BYTES]; \
  STile t;                                                                   \
  t.img = lds;                                                               \
  t.pmax = reinterpret_cast<float*>(lds + 2 * NS_PLANE);                     \
  t.r = threadIdx.x & 31;                                                    \
  t.h = (threadIdx.x >> 5) & 1;                                              \
  t.nb = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);                   \
  const int row = blockIdx.x * 32 + t.r;                                     \
  const int rc = min(row, p.N - 1);                                          \
  const bool live = row < p.N;

// equiv_update + energy update + the next layer's message_nodepart (or the first two linears of the energy head)
__global__ void __launch_bounds__(256, NS_WG_PER_CU) node_fwd_split_kernel(const NodeFwdArgs p, const NodeImages im) {
  NS_DBG_INIT()
  NS_TILE_SETUP()
  WFrag wf;
  load_wimg(wf, t, im.Wu);
  float upd[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) upd[k] = 0.f;
  float a[16];
#if NS_MERGE3
  {
    STile t1 = t, t2 = t;
    t1.img = lds + NS_LDS_BYTES;
    t1.pmax = reinterpret_cast<float*>(lds + NS_LDS_BYTES + 2 * NS_PLANE);
    t2.img = lds + 2 * NS_LDS_BYTES;
    t2.pmax = reinterpret_cast<float*>(lds + 2 * NS_LDS_BYTES + 2 * NS_PLANE);
    float x0[16], x1[16], x2[16], qv[16];
    sblk_load(x0, p.f, ((size_t)rc * 3 + 0) * NF, t);
    sblk_load(x1, p.f, ((size_t)rc * 3 + 1) * NF, t);
    sblk_load(x2, p.f, ((size_t)rc * 3 + 2) * NF, t);
    tile_publish(x0, t);
    tile_publish(x1, t1);
    tile_publish(x2, t2);
    __syncthreads();
    const float inv0 = tile_commit(x0, t), inv1 = tile_commit(x1, t1), inv2 = tile_commit(x2, t2);
    __syncthreads();
    sblk_load(a, p.a_mid, (size_t)rc * NF, t);
    tile_gemm_s(qv, t, wf, inv0);
    if (live) sblk_store(qv, p.q, ((size_t)row * 3 + 0) * NF, t);
#pragma unroll
    for (int k = 0; k < 16; ++k) upd[k] = x0[k] * qv[k];
    tile_gemm_s(qv, t1, wf, inv1);
    if (live) sblk_store(qv, p.q, ((size_t)row * 3 + 1) * NF, t);
#pragma unroll
    for (int k = 0; k < 16; ++k) upd[k] = fmaf(x1[k], qv[k], upd[k]);
    tile_gemm_s(qv, t2, wf, inv2);
    if (p.W0) load_wimg(wf, t, im.W0);
    if (live) sblk_store(qv, p.q, ((size_t)row * 3 + 2) * NF, t);
#pragma unroll
    for (int k = 0; k < 16; ++k) upd[k] = fmaf(x2[k], qv[k], upd[k]);
  }
#else
  float xa[16], xb[16];
  sblk_load(xa, p.f, ((size_t)rc * 3 + 0) * NF, t);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float (&cur)[16] = (c & 1) ? xb : xa;
    float (&nxt)[16] = (c & 1) ? xa : xb;
    float qv[16];
    tile_publish(cur, t);
    __syncthreads();            // every wave is done reading the previous tile; the row maxima are visible
    const float inv = tile_commit(cur, t);
    __syncthreads();
    if (c < 2)
      sblk_load(nxt, p.f, ((size_t)rc * 3 + c + 1) * NF, t);
    else
      sblk_load(a, p.a_mid, (size_t)rc * NF, t);
    NS_DBG_STAMP()
    tile_gemm_s(qv, t, wf, inv);
    NS_DBG_STAMP()
    if (c == 2 && p.W0) load_wimg(wf, t, im.W0);
#ifndef NS_ABL_NO_Q   // tooling (wrong forces): the update without its q = f W_u^T round trip -- what recomputing q in the adjoint could save at most
    if (live) sblk_store(qv, p.q, ((size_t)row * 3 + c) * NF, t);
#endif
#pragma unroll
    for (int k = 0; k < 16; ++k) upd[k] = fmaf(cur[k], qv[k], upd[k]);
  }
#endif
#pragma unroll
  for (int k = 0; k < 16; ++k) a[k] += upd[k];
  if (live) sblk_store(a, p.a_out, (size_t)row * NF, t);
  if (!p.W0) return;

  // message_nodepart of the next layer
  tile_publish(a, t);
  __syncthreads();
  float inv = tile_commit(a, t);
  __syncthreads();
  float hn[16], b0v[16], b2v[16];
  sblk_load(b0v, p.b0, 0, t);
  NS_DBG_STAMP()
  tile_gemm_s(hn, t, wf, inv);
  NS_DBG_STAMP()
  load_wimg(wf, t, im.W2);
#pragma unroll
  for (int k = 0; k < 16; ++k) hn[k] += b0v[k];
  if (live) sblk_store(hn, p.hn, (size_t)row * NF, t);
#pragma unroll
  for (int k = 0; k < 16; ++k) hn[k] = silu_f(hn[k]);
  tile_publish(hn, t);
  __syncthreads();
  inv = tile_commit(hn, t);
  __syncthreads();
  sblk_load(b2v, p.b2, 0, t);
  float m[16];
  NS_DBG_STAMP()
  tile_gemm_s(m, t, wf, inv);
  NS_DBG_STAMP()
#pragma unroll
  for (int k = 0; k < 16; ++k) m[k] += b2v[k];
  if (live) sblk_store(m, p.m, (size_t)row * NF, t);
  NS_DBG_PRINT("node_fwd")
}

// adjoint of the upper node MLP / head, then of the lower layer's update (see node128.hip:node_bwd_kernel)
#ifndef NS_WG_PER_CU_BWD
#define NS_WG_PER_CU_BWD NS_WG_PER_CU
#endif
__global__ void __launch_bounds__(256, NS_WG_PER_CU_BWD) node_bwd_split_kernel(const NodeBwdArgs p, const NodeImages im) {
  NS_TILE_SETUP()
  float ga[16];
  WFrag wf;
  if (p.W2T) {
    load_wimg(wf, t, im.W2T);
    float x[16], hpre[16], g[16];
    sblk_load(x, p.g_top, (size_t)rc * NF, t);
    sblk_load(hpre, p.h_top, (size_t)rc * NF, t);
    tile_publish(x, t);
    __syncthreads();
    float inv = tile_commit(x, t);
    __syncthreads();
    tile_gemm_s(g, t, wf, inv);
    load_wimg(wf, t, im.W0T);
    if (p.T && live) sblk_store(g, p.T, (size_t)row * NF, t);
#pragma unroll
    for (int k = 0; k < 16; ++k) g[k] *= dsilu_f(hpre[k]);
    tile_publish(g, t);
    __syncthreads();
    inv = tile_commit(g, t);
    __syncthreads();
    tile_gemm_s(ga, t, wf, inv);
    if (p.WuT) load_wimg(wf, t, im.WuT);
    if (p.acc_ga) {
      float old[16];
      sblk_load(old, p.g_a_in ? p.g_a_in : p.g_a, (size_t)rc * NF, t);
#pragma unroll
      for (int k = 0; k < 16; ++k) ga[k] += old[k];
    }
    if (live) sblk_store(ga, p.g_a, (size_t)row * NF, t);
  } else {
    sblk_load(ga, p.g_a, (size_t)rc * NF, t);
    if (p.WuT) load_wimg(wf, t, im.WuT);
  }
  if (!p.WuT) return;

  // adjoint of the lower layer's update:  gf_k = G_f,k + g_a * q_k + (g_a * f'_k) W_u
#if NS_MERGE3
  {
    STile t1 = t, t2 = t;
    t1.img = lds + NS_LDS_BYTES;
    t1.pmax = reinterpret_cast<float*>(lds + NS_LDS_BYTES + 2 * NS_PLANE);
    t2.img = lds + 2 * NS_LDS_BYTES;
    t2.pmax = reinterpret_cast<float*>(lds + 2 * NS_LDS_BYTES + 2 * NS_PLANE);
    float x0[16], x1[16], x2[16];
    sblk_load(x0, p.f, ((size_t)rc * 3 + 0) * NF, t);
    sblk_load(x1, p.f, ((size_t)rc * 3 + 1) * NF, t);
    sblk_load(x2, p.f, ((size_t)rc * 3 + 2) * NF, t);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      x0[k] *= ga[k];
      x1[k] *= ga[k];
      x2[k] *= ga[k];
    }
    tile_publish(x0, t);
    tile_publish(x1, t1);
    tile_publish(x2, t2);
    __syncthreads();
    const float inv0 = tile_commit(x0, t), inv1 = tile_commit(x1, t1), inv2 = tile_commit(x2, t2);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float out[16], qv[16], gin[16];
      sblk_load(qv, p.q, ((size_t)rc * 3 + c) * NF, t);
      if (p.G_f) sblk_load(gin, p.G_f, ((size_t)rc * 3 + c) * NF, t);
      tile_gemm_s(out, c == 0 ? t : (c == 1 ? t1 : t2), wf, c == 0 ? inv0 : (c == 1 ? inv1 : inv2));
#pragma unroll
      for (int k = 0; k < 16; ++k) out[k] = fmaf(ga[k], qv[k], out[k]);
      if (p.G_f) {
#pragma unroll
        for (int k = 0; k < 16; ++k) out[k] += gin[k];
      }
      if (live) sblk_store(out, p.gf, ((size_t)row * 3 + c) * NF, t);
    }
  }
#else
  float fa[16], fb[16];
  sblk_load(fa, p.f, ((size_t)rc * 3 + 0) * NF, t);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float (&cur)[16] = (c & 1) ? fb : fa;
    float (&nxt)[16] = (c & 1) ? fa : fb;
    float out[16], qv[16], gin[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) cur[k] *= ga[k];
    tile_publish(cur, t);
    __syncthreads();
    const float inv = tile_commit(cur, t);
    __syncthreads();
    if (c < 2) sblk_load(nxt, p.f, ((size_t)rc * 3 + c + 1) * NF, t);
#ifdef NS_ABL_NO_Q
#pragma unroll
    for (int k = 0; k < 16; ++k) qv[k] = cur[k];
#else
    sblk_load(qv, p.q, ((size_t)rc * 3 + c) * NF, t);
#endif
    if (p.G_f) sblk_load(gin, p.G_f, ((size_t)rc * 3 + c) * NF, t);
    tile_gemm_s(out, t, wf, inv);
#pragma unroll
    for (int k = 0; k < 16; ++k) out[k] = fmaf(ga[k], qv[k], out[k]);
    if (p.G_f) {
#pragma unroll
      for (int k = 0; k < 16; ++k) out[k] += gin[k];
    }
    if (live) sblk_store(out, p.gf, ((size_t)row * 3 + c) * NF, t);
  }
#endif
}

// Tangent of node_fwd (training sweep 3): dq_k = df_k W_u^T;  da_out = da_mid + sum_k (df_k q_k + f_k dq_k);
// T = da_out W0^T (tangent of hn);  Y = (T silu'(hn)) W2^T (tangent of the next m / of e2).  Images: Wu, W0, W2.
__global__ void __launch_bounds__(256, 2) node_tan_fwd_split_kernel(const NodeTanFwdArgs p, const NodeImages im) {
  NS_TILE_SETUP()
  WFrag wf;
  load_wimg(wf, t, im.Wu);
  float upd[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) upd[k] = 0.f;
  float xa[16], xb[16], a[16];
  sblk_load(xa, p.df, ((size_t)rc * 3 + 0) * NF, t);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float (&cur)[16] = (c & 1) ? xb : xa;
    float (&nxt)[16] = (c & 1) ? xa : xb;
    float dqv[16], fv[16], qv[16];
    tile_publish(cur, t);
    __syncthreads();
    const float inv = tile_commit(cur, t);
    __syncthreads();
    if (c < 2)
      sblk_load(nxt, p.df, ((size_t)rc * 3 + c + 1) * NF, t);
    else
      sblk_load(a, p.da_mid, (size_t)rc * NF, t);
    sblk_load(fv, p.f, ((size_t)rc * 3 + c) * NF, t);
    sblk_load(qv, p.q, ((size_t)rc * 3 + c) * NF, t);
    tile_gemm_s(dqv, t, wf, inv);
    if (c == 2) load_wimg(wf, t, im.W0);
    if (live) sblk_store(dqv, p.dq, ((size_t)row * 3 + c) * NF, t);
#pragma unroll
    for (int k = 0; k < 16; ++k) upd[k] = fmaf(cur[k], qv[k], fmaf(fv[k], dqv[k], upd[k]));
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) a[k] += upd[k];
  if (live) sblk_store(a, p.da_out, (size_t)row * NF, t);

  tile_publish(a, t);
  __syncthreads();
  float inv = tile_commit(a, t);
  __syncthreads();
  float hn[16], tv[16];
  sblk_load(hn, p.hn, (size_t)rc * NF, t);
  tile_gemm_s(tv, t, wf, inv);
  load_wimg(wf, t, im.W2);
  if (live) sblk_store(tv, p.T, (size_t)row * NF, t);
#pragma unroll
  for (int k = 0; k < 16; ++k) tv[k] *= dsilu_f(hn[k]);
  tile_publish(tv, t);
  __syncthreads();
  inv = tile_commit(tv, t);
  __syncthreads();
  float y[16];
  tile_gemm_s(y, t, wf, inv);
  if (live) sblk_store(y, p.Y, (size_t)row * NF, t);
}

// Tangent of node_bwd (training sweep 4).  Part A, tangent of the upper node-MLP / head adjoint (images W2T, W0T):
//   dT = g_top W2;  G = dT silu'(h) + t2 silu''(h) hd (stored);  dGA (+)= G W0.
// Part B, tangent of the lower layer's update adjoint (image WuT), with the dGA part A just produced:
//   gq_k = GA f_k;  dgq_k = dGA f_k + GA df_k;  dgf_k = dG_f,k + dGA q_k + GA dq_k + dgq_k W_u.
__global__ void __launch_bounds__(256, 2) node_tan_bwd_split_kernel(const NodeTanBwdArgs p, const NodeImages im) {
  NS_TILE_SETUP()
  float dga[16];
  WFrag wf;
  if (p.g_top) {
    load_wimg(wf, t, im.W2T);
    float x[16], hpre[16], g[16];
    sblk_load(x, p.g_top, (size_t)rc * NF, t);
    sblk_load(hpre, p.h_top, (size_t)rc * NF, t);
    tile_publish(x, t);
    __syncthreads();
    float inv = tile_commit(x, t);
    __syncthreads();
    tile_gemm_s(g, t, wf, inv);
    load_wimg(wf, t, im.W0T);
    if (p.t2_top) {   // (uniform)
      float t2[16], hd[16];
      sblk_load(t2, p.t2_top, (size_t)rc * NF, t);
      sblk_load(hd, p.hd_top, (size_t)rc * NF, t);
#pragma unroll
      for (int k = 0; k < 16; ++k) g[k] = fmaf(g[k], dsilu_f(hpre[k]), t2[k] * d2silu_f(hpre[k]) * hd[k]);
    } else {
#pragma unroll
      for (int k = 0; k < 16; ++k) g[k] *= dsilu_f(hpre[k]);
    }
    if (live) sblk_store(g, p.G, (size_t)row * NF, t);
    tile_publish(g, t);
    __syncthreads();
    inv = tile_commit(g, t);
    __syncthreads();
    tile_gemm_s(dga, t, wf, inv);
    if (p.f) load_wimg(wf, t, im.WuT);
    if (p.acc_dga) {
      float old[16];
      sblk_load(old, p.dga, (size_t)rc * NF, t);
#pragma unroll
      for (int k = 0; k < 16; ++k) dga[k] += old[k];
    }
    if (live) sblk_store(dga, p.dga, (size_t)row * NF, t);
  } else {
    sblk_load(dga, p.dga, (size_t)rc * NF, t);
    if (p.f) load_wimg(wf, t, im.WuT);
  }
  if (!p.f) return;

  float ga[16];
  sblk_load(ga, p.ga, (size_t)rc * NF, t);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const size_t off = ((size_t)rc * 3 + c) * NF, ooff = ((size_t)row * 3 + c) * NF;
    float fv[16], dfv[16], v[16], out[16];
    sblk_load(fv, p.f, off, t);
    sblk_load(dfv, p.df, off, t);
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = ga[k] * fv[k];
    if (live) sblk_store(v, p.gq, ooff, t);
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = fmaf(dga[k], fv[k], ga[k] * dfv[k]);
    if (live) sblk_store(v, p.dgq, ooff, t);
    tile_publish(v, t);
    __syncthreads();
    const float inv = tile_commit(v, t);
    __syncthreads();
    sblk_load(fv, p.q, off, t);       // (fv, dfv reused for q, dq: consumed after the GEMM)
    sblk_load(dfv, p.dq, off, t);
    tile_gemm_s(out, t, wf, inv);
#pragma unroll
    for (int k = 0; k < 16; ++k) out[k] += fmaf(dga[k], fv[k], ga[k] * dfv[k]);
    if (p.dgf_in) {
      float gin[16];
      sblk_load(gin, p.dgf_in, off, t);
#pragma unroll
      for (int k = 0; k < 16; ++k) out[k] += gin[k];
    }
    if (live) sblk_store(out, p.dgf, ooff, t);
  }
}

int launch_node_tan_fwd_split(const NodeTanFwdArgs& a, const NodeImages& im, hipStream_t s) {
  if (a.N <= 0) return 0;
  ScopedTimer t0(TC_LIN, s);
  node_tan_fwd_split_kernel<<<cdiv(a.N, 32), 256, 0, s>>>(a, im);
  LAUNCH_CHECK();
  return 0;
}
int launch_node_tan_bwd_split(const NodeTanBwdArgs& a, const NodeImages& im, hipStream_t s) {
  if (a.N <= 0) return 0;
  ScopedTimer t0(TC_LIN, s);
  node_tan_bwd_split_kernel<<<cdiv(a.N, 32), 256, 0, s>>>(a, im);
  LAUNCH_CHECK();
  return 0;
}

int launch_node_fwd_split(const NodeFwdArgs& a, const NodeImages& im, hipStream_t s) {
  if (a.N <= 0) return 0;
  ScopedTimer t0(TC_LIN, s);
  ScopedTimer t1(TC_LIN1, s);
  node_fwd_split_kernel<<<cdiv(a.N, 32), 256, 0, s>>>(a, im);
  LAUNCH_CHECK();
  return 0;
}
int launch_node_bwd_split(const NodeBwdArgs& a, const NodeImages& im, hipStream_t s) {
  if (a.N <= 0) return 0;
  ScopedTimer t0(TC_LIN, s);
  ScopedTimer t1(TC_LIN1, s);
  node_bwd_split_kernel<<<cdiv(a.N, 32), 256, 0, s>>>(a, im);
  LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Row-local form of the fused edge / node MLP (node128.hip: mlp128_wide_kernel, same MlpArgs contract, SiLU) with split-f16
// products: the weights come as prepared images (MlpArgs::W1_img / W2_img).  The single-molecule / MD-loop and small-batch
// training regimes are chains of these launches; a GEMM stage is 24 MFMAs of 32 cycles instead of 64 of 64.
// ---------------------------------------------------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ void mlp_wide_split_body(const MlpArgs& p, const bool accum, STile& t) {
  const int M = mlp_rows(p);
  if ((int)blockIdx.x * 32 >= M) return;   // (block-uniform: a tile beyond a device-side row count; never taken otherwise)
  const int row = blockIdx.x * 32 + t.r;
  const int rc = min(row, M - 1);
  const bool live = row < M;

  // bf16 compute mode: what the IMAGES say (nnhip_weight_images_bf16 wrote them; the two images of an MLP share a format)
  const bool bf = wimg_is_bf16(p.W1_img);
  WFrag wf;
  load_wimg(wf, t, p.W1_img);
  float x[16], hv[16], hin[16];
  sblk_load(x, p.X, (size_t)rc * p.ldx, t);
  if (MODE != MODE_FWD) sblk_load(hin, p.H, (size_t)rc * p.ldh, t);
  float inv = 1.0f;
  if (bf) {
    __syncthreads();
    tile_commit_bf(x, t);
    __syncthreads();
    tile_gemm_bf(hv, t, wf);
  } else {
    tile_publish(x, t);
    __syncthreads();
    inv = tile_commit(x, t);
    __syncthreads();
    tile_gemm_s(hv, t, wf, inv);
  }
  load_wimg(wf, t, p.W2_img);
  if (MODE == MODE_FWD) {
    if (p.b1) {
      float b[16];
      sblk_load(b, p.b1, 0, t);
#pragma unroll
      for (int k = 0; k < 16; ++k) hv[k] += b[k];
    }
    if (live) sblk_store(hv, p.H, (size_t)row * p.ldh, t);
#pragma unroll
    for (int k = 0; k < 16; ++k) hv[k] = silu_f(hv[k]);
  } else if (MODE == MODE_TAN2) {
    float t2[16], hd[16];
    sblk_load(t2, p.T2, (size_t)rc * p.ldh, t);
    sblk_load(hd, p.Hd, (size_t)rc * p.ldh, t);
#pragma unroll
    for (int k = 0; k < 16; ++k) hv[k] = fmaf(hv[k], dsilu_f(hin[k]), t2[k] * d2silu_f(hin[k]) * hd[k]);
    if (live) sblk_store(hv, p.G, (size_t)row * p.ldh, t);
  } else {
    if (MODE == MODE_TAN && live) sblk_store(hv, p.T, (size_t)row * p.ldh, t);
#pragma unroll
    for (int k = 0; k < 16; ++k) hv[k] *= dsilu_f(hin[k]);
  }
  float y[16];
  if (bf) {
    __syncthreads();
    tile_commit_bf(hv, t);
    __syncthreads();
    tile_gemm_bf(y, t, wf);
  } else {
    tile_publish(hv, t);
    __syncthreads();
    inv = tile_commit(hv, t);
    __syncthreads();
    tile_gemm_s(y, t, wf, inv);
  }
  if (MODE == MODE_FWD && p.b2) {
    float b[16];
    sblk_load(b, p.b2, 0, t);
#pragma unroll
    for (int k = 0; k < 16; ++k) y[k] += b[k];
  }
  if (accum) {   // uniform
    float yold[16];
    sblk_load(yold, p.Y, (size_t)rc * p.ldy, t);
#pragma unroll
    for (int k = 0; k < 16; ++k) y[k] += yold[k];
  }
  if (live) sblk_store(y, p.Y, (size_t)row * p.ldy, t);
}

#define NS_STILE_SETUP()                                                     \
  __shared__ __attribute__((aligned(16))) char lds[NS_LDS_BYTES];            \
  STile t;                                                                   \
  t.img = lds;                                                               \
  t.pmax = reinterpret_cast<float*>(lds + 2 * NS_PLANE);                     \
  t.r = threadIdx.x & 31;                                                    \
  t.h = (threadIdx.x >> 5) & 1;                                              \
  t.nb = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

template <int MODE, bool ACCUM>
__global__ void __launch_bounds__(256) mlp128_wide_split_kernel(const MlpArgs p) {
  NS_STILE_SETUP()
  mlp_wide_split_body<MODE>(p, ACCUM, t);
}
// two MLPs over the same rows in one launch (node128.hip:mlp128_wide_pair_kernel)
template <int MODE, bool PAR>
__global__ void __launch_bounds__(256) mlp128_wide_pair_split_kernel(const MlpPair P) {
  NS_STILE_SETUP()
  if (PAR) {
    if (blockIdx.y == 0)
      mlp_wide_split_body<MODE>(P.a[0], false, t);
    else
      mlp_wide_split_body<MODE>(P.a[1], false, t);
  } else {
    mlp_wide_split_body<MODE>(P.a[0], false, t);
    __syncthreads();
    mlp_wide_split_body<MODE>(P.a[1], P.accum[1] != 0, t);
  }
}

int launch_mlp_wide_split(int mode, bool accum, const MlpArgs& a, hipStream_t s) {
  const int n_tiles = cdiv(a.M, 32);
#define WIDE_S(M_, A_)                                                       \
  if (mode == M_ && accum == A_) {                                           \
    mlp128_wide_split_kernel<M_, A_><<<n_tiles, 256, 0, s>>>(a);             \
    LAUNCH_CHECK();                                                          \
    return 0;                                                                \
  }
  WIDE_S(MODE_FWD, false)
  WIDE_S(MODE_BWD, false)
  WIDE_S(MODE_BWD, true)
  WIDE_S(MODE_TAN, false)
  WIDE_S(MODE_TAN, true)
  WIDE_S(MODE_TAN2, false)
  WIDE_S(MODE_TAN2, true)
#undef WIDE_S
  return NNHIP_E_INVALID;
}
int launch_mlp_wide_pair_split(int mode, const MlpPair& P, hipStream_t s) {
  const int n_tiles = cdiv(P.a[0].M, 32);
  const bool par = !P.accum[1];
#define WIDE_PAIR_S(M_)                                                                  \
  if (mode == M_) {                                                                      \
    if (par)                                                                             \
      mlp128_wide_pair_split_kernel<M_, true><<<dim3(n_tiles, 2), 256, 0, s>>>(P);       \
    else                                                                                 \
      mlp128_wide_pair_split_kernel<M_, false><<<n_tiles, 256, 0, s>>>(P);               \
    LAUNCH_CHECK();                                                                      \
    return 0;                                                                            \
  }
  WIDE_PAIR_S(MODE_FWD)
  WIDE_PAIR_S(MODE_BWD)
  WIDE_PAIR_S(MODE_TAN)
  WIDE_PAIR_S(MODE_TAN2)
#undef WIDE_PAIR_S
  return NNHIP_E_INVALID;
}

extern "C" size_t nnhip_weight_image_bytes(void) { return ((size_t)WIMG_BYTES + 255) & ~(size_t)255; }
static int weight_images_impl(const float* const* src, void* const* images, int32_t count, void* stream, int fmt);
extern "C" int nnhip_weight_images(const float* const* src, void* const* images, int32_t count, void* stream) {
  return weight_images_impl(src, images, count, stream, WIMG_FMT_SPLIT_F16);
}
extern "C" int nnhip_weight_images_bf16(const float* const* src, void* const* images, int32_t count, void* stream) {
  return weight_images_impl(src, images, count, stream, WIMG_FMT_BF16);
}
static int weight_images_impl(const float* const* src, void* const* images, int32_t count, void* stream, int fmt) {
  if (!src || !images || count < 0) {
    nnhip_set_error("nnhip_weight_images: bad arguments");
    return NNHIP_E_INVALID;
  }
  for (int k = 0; k < count; ++k)
    if (!src[k] || !images[k] || ((uintptr_t)images[k] & 15)) {
      nnhip_set_error("nnhip_weight_images: matrix %d: null or misaligned pointer", k);
      return NNHIP_E_INVALID;
    }
  return count ? launch_weight_images(src, reinterpret_cast<char* const*>(images), count, (hipStream_t)stream, fmt) : NNHIP_OK;
}

// One dense linear Y (+)= X W^T, row-local, split-f16 products (node128.hip:lin128_wide_kernel with a weight image)
template <bool ACC>
__global__ void __launch_bounds__(256)
lin128_wide_split_kernel(const float* __restrict__ X, int ldx, const char* __restrict__ img, float* __restrict__ Y, int ldy, int M) {
  NS_STILE_SETUP()
  const int row = blockIdx.x * 32 + t.r;
  const int rc = min(row, M - 1);
  WFrag wf;
  load_wimg(wf, t, img);
  float x[16], y[16];
  sblk_load(x, X, (size_t)rc * ldx, t);
  tile_publish(x, t);
  __syncthreads();
  const float inv = tile_commit(x, t);
  __syncthreads();
  if (ACC) sblk_load(x, Y, (size_t)rc * ldy, t);
  tile_gemm_s(y, t, wf, inv);
  if (ACC) {
#pragma unroll
    for (int k = 0; k < 16; ++k) y[k] += x[k];
  }
  if (row < M) sblk_store(y, Y, (size_t)row * ldy, t);
}
int launch_lin_wide_split(const float* X, int ldx, const char* img, float* Y, int ldy, int M, bool acc, hipStream_t s) {
  if (M <= 0) return 0;
  ScopedTimer t0(TC_LIN, s);
  if (acc)
    lin128_wide_split_kernel<true><<<cdiv(M, 32), 256, 0, s>>>(X, ldx, img, Y, ldy, M);
  else
    lin128_wide_split_kernel<false><<<cdiv(M, 32), 256, 0, s>>>(X, ldx, img, Y, ldy, M);
  LAUNCH_CHECK();
  return 0;
}
