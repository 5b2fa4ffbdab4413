// Row-local fused node kernels (gfx950, exact fp32 MFMA): everything that happens to the node tensors BETWEEN two
// edge phases of the path, in one launch.
//
// Forward boundary after the edge phase of layer l (newtonnet/models/newtonnet.py:229-231 and, for the next layer,
// :181-185,209):
//     q_k   = f'_k W_u^T                    k = 0,1,2          (equiv_update)
//     a_out = a_mid + sum_k f'_k * q_k                           (energy update)
//     hn    = a_out W_0^T + b_0 ;  m = silu(hn) W_2^T + b_2      (message_nodepart of layer l+1, if any)
// Reverse boundary (the adjoint of the same operations, walking down):
//     g_hn  = (g_m W_2) * silu'(hn) ;  g_a (+)= g_hn W_0          (message_nodepart adjoint of the upper layer, or the
//                                                                  energy-head adjoint at the top)
//     gf_k  = G_f,k + g_a * q_k + (g_a * f'_k) W_u                (update adjoint of the lower layer)
//
// All of it is row-local (no atom talks to another), but as separate launches it was ~20 latency-bound kernels per
// step (a 0.7 GFLOP GEMM takes ~20 us whatever the kernel: one 256-MFMA chain per wave).  Here one 4-wave workgroup
// owns a 32-atom tile and walks the whole chain; wave w computes output column block w of every GEMM (64-MFMA chains),
// activations pass from one GEMM to the next through a 16.5 KiB LDS tile, and the weights come straight from L2 as
// the MFMA A operand (transposed formulation, as in mlp128.hip: D^T[feature][atom] = W X^T; every wave of the grid
// with the same w reads the same 16 KiB of each matrix).
#include <string.h>

#include "common.h"

#ifndef NODE_WG_PER_CU
#define NODE_WG_PER_CU 2   // launch-bounds hint (min waves per SIMD = workgroups per CU for 256-thread workgroups); tooling: 3
#endif
#define NT_LD 132                       // LDS tile row pitch (floats): conflict-free ds_read_b128 / ds_write_b128
#define NODE_LDS_FLOATS (32 * NT_LD)

struct Tile {
  float* xs;  // LDS [32][NT_LD]
  int r, h, nb;
};

// Weight fragment of column block `nb`: rows nb*32 .. nb*32+31 of W straight from global (L2), k-slots 8k + 4h + {0..3}.
// Fetched one GEMM AHEAD (right after the previous GEMM's MFMAs have consumed the registers), so the L2 round trip hides
// under the epilogue / LDS hand-over / barrier of the previous stage.
__device__ __forceinline__ void load_w(float4 (&wf)[16], const Tile& t, const float* __restrict__ W) {
  const float4* wp = reinterpret_cast<const float4*>(W + (size_t)(t.nb * 32 + t.r) * NF + 4 * t.h);
#pragma unroll
  for (int k = 0; k < 16; ++k) wf[k] = wp[2 * k];
  __builtin_amdgcn_sched_barrier(0);   // keep the 16 loads together and ahead of what follows
}

// block `nb` of D^T = W . X^T for the 32 rows in the LDS tile
__device__ __forceinline__ f32x16 tile_gemm(const Tile& t, const float4 (&wf)[16]) {
  f32x16 acc;
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.f;
  const float* xr = t.xs + t.r * NT_LD + 4 * t.h;
  // B fragments come from LDS two k-groups ahead of their use ("1 ds_read, 4 MFMA" pinned with sched_group_barrier; left
  // alone the compiler sinks each read to just before its MFMAs and the chain stalls for the LDS latency 16 times)
  float4 x0 = *reinterpret_cast<const float4*>(xr), x1 = *reinterpret_cast<const float4*>(xr + 8);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    float4 x2;
    if (k < 14) x2 = *reinterpret_cast<const float4*>(xr + 8 * (k + 2));
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[k].x, x0.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[k].y, x0.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[k].z, x0.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[k].w, x0.w, acc, 0, 0, 0);
    x0 = x1;
    if (k < 14) x1 = x2;
    if (k < 14) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
  return acc;
}

// this lane's 16 values of column block nb (features nb*32 + (k&3) + 8 (k>>2) + 4h) of row `row`
__device__ __forceinline__ void blk_load(float (&v)[16], const float* __restrict__ base, size_t row_off, const Tile& t) {
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
#ifndef NODE_NT
#define NODE_NT 0   // streaming stores of the tensors kept only for the reverse sweep (q, hn): no measurable effect, off
#endif
template <bool NT = false>
__device__ __forceinline__ void blk_store(const float (&v)[16], float* __restrict__ base, size_t row_off, const Tile& t) {
  float4* p = reinterpret_cast<float4*>(base + row_off + t.nb * 32 + 4 * t.h);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 w = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    if (NT)
      st4_nt(reinterpret_cast<float*>(p + 2 * q), w);
    else
      p[2 * q] = w;
  }
}
__device__ __forceinline__ void blk_to_tile(const float (&v)[16], const Tile& t) {
  float* p = t.xs + t.r * NT_LD + t.nb * 32 + 4 * t.h;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    *reinterpret_cast<float4*>(p + 8 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}
__device__ __forceinline__ void acc_to(float (&v)[16], const f32x16& a) {
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = a[k];
}

__global__ void __launch_bounds__(256, NODE_WG_PER_CU) node_fwd_kernel(const NodeFwdArgs p) {
  __shared__ __attribute__((aligned(16))) float xs[NODE_LDS_FLOATS];
  Tile t;
  t.xs = xs;
  t.r = threadIdx.x & 31;
  t.h = (threadIdx.x >> 5) & 1;
  t.nb = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row = blockIdx.x * 32 + t.r;
  const int rc = min(row, p.N - 1);   // clamped row for loads; stores are predicated
  const bool live = row < p.N;

  float4 wf[16];
  load_w(wf, t, p.Wu);                 // one fetch serves the three components
  float upd[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) upd[k] = 0.f;
  // Operands of the NEXT stage are requested before each GEMM (the sched_barrier at the top of tile_gemm keeps them
  // ahead of its MFMAs): all workgroups of the grid start together and walk the chain in lockstep, so a load issued only
  // when it is needed is exposed on every CU at the same time.
  float xa[16], xb[16], a[16];
  blk_load(xa, p.f, ((size_t)rc * 3 + 0) * NF, t);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float (&cur)[16] = (c & 1) ? xb : xa;
    float (&nxt)[16] = (c & 1) ? xa : xb;
    float qv[16];
    if (c) __syncthreads();            // every wave is done reading the previous tile
    blk_to_tile(cur, t);
    __syncthreads();
    if (c < 2)
      blk_load(nxt, p.f, ((size_t)rc * 3 + c + 1) * NF, t);
    else
      blk_load(a, p.a_mid, (size_t)rc * NF, t);
    acc_to(qv, tile_gemm(t, wf));
    if (c == 2 && p.W0) load_w(wf, t, p.W0);
    if (live) blk_store<NODE_NT != 0>(qv, p.q, ((size_t)row * 3 + c) * NF, t);
#pragma unroll
    for (int k = 0; k < 16; ++k) upd[k] = fmaf(cur[k], qv[k], upd[k]);
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) a[k] += upd[k];
  if (live) blk_store(a, p.a_out, (size_t)row * NF, t);
  if (!p.W0) return;

  // message_nodepart of the next layer
  __syncthreads();
  blk_to_tile(a, t);
  __syncthreads();
  float hn[16], b0v[16], b2v[16];
  blk_load(b0v, p.b0, 0, t);
  acc_to(hn, tile_gemm(t, wf));
  load_w(wf, t, p.W2);
#pragma unroll
  for (int k = 0; k < 16; ++k) hn[k] += b0v[k];
  if (live) blk_store<NODE_NT != 0>(hn, p.hn, (size_t)row * NF, t);
  NN_ACT_BLOCK(hn, 16, p.act);
  __syncthreads();
  blk_to_tile(hn, t);
  __syncthreads();
  blk_load(b2v, p.b2, 0, t);
  float m[16];
  acc_to(m, tile_gemm(t, wf));
#pragma unroll
  for (int k = 0; k < 16; ++k) m[k] += b2v[k];
  if (live) blk_store(m, p.m, (size_t)row * NF, t);
}

__global__ void __launch_bounds__(256, NODE_WG_PER_CU) node_bwd_kernel(const NodeBwdArgs p) {
  __shared__ __attribute__((aligned(16))) float xs[NODE_LDS_FLOATS];
  Tile t;
  t.xs = xs;
  t.r = threadIdx.x & 31;
  t.h = (threadIdx.x >> 5) & 1;
  t.nb = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row = blockIdx.x * 32 + t.r;
  const int rc = min(row, p.N - 1);
  const bool live = row < p.N;

  float ga[16];
  float4 wf[16];
  if (p.W2T) {
    // adjoint of the upper two-layer node MLP / head:  g_hn = (g_top W2) * silu'(h_top);  g_a (+)= g_hn W0
    load_w(wf, t, p.W2T);
    float x[16], hpre[16], g[16];
    blk_load(x, p.g_top, (size_t)rc * NF, t);
    blk_load(hpre, p.h_top, (size_t)rc * NF, t);
    blk_to_tile(x, t);
    __syncthreads();
    acc_to(g, tile_gemm(t, wf));
    load_w(wf, t, p.W0T);
    NN_DACT_MUL_BLOCK(g, hpre, 16, p.act);
    __syncthreads();
    blk_to_tile(g, t);
    __syncthreads();
    acc_to(ga, tile_gemm(t, wf));
    if (p.WuT) load_w(wf, t, p.WuT);
    if (p.acc_ga) {
      float old[16];
      blk_load(old, p.g_a, (size_t)rc * NF, t);
#pragma unroll
      for (int k = 0; k < 16; ++k) ga[k] += old[k];
    }
    if (live) blk_store(ga, p.g_a, (size_t)row * NF, t);
  } else {
    blk_load(ga, p.g_a, (size_t)rc * NF, t);
    if (p.WuT) load_w(wf, t, p.WuT);
  }
  if (!p.WuT) return;

  // adjoint of the lower layer's update:  gf_k = G_f,k + g_a * q_k + (g_a * f'_k) W_u
  // (the next component's f, q and G_f are requested before each GEMM, see node_fwd_kernel)
  float fa[16], fb[16];
  blk_load(fa, p.f, ((size_t)rc * 3 + 0) * NF, t);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float (&cur)[16] = (c & 1) ? fb : fa;
    float (&nxt)[16] = (c & 1) ? fa : fb;
    float out[16], qv[16], gin[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) cur[k] *= ga[k];
    __syncthreads();
    blk_to_tile(cur, t);
    __syncthreads();
    if (c < 2) blk_load(nxt, p.f, ((size_t)rc * 3 + c + 1) * NF, t);
    blk_load(qv, p.q, ((size_t)rc * 3 + c) * NF, t);              // consumed after the GEMM: hidden under it
    if (p.G_f) blk_load(gin, p.G_f, ((size_t)rc * 3 + c) * NF, t);
    acc_to(out, tile_gemm(t, wf));
#pragma unroll
    for (int k = 0; k < 16; ++k) out[k] = fmaf(ga[k], qv[k], out[k]);
    if (p.G_f) {
#pragma unroll
      for (int k = 0; k < 16; ++k) out[k] += gin[k];
    }
    if (live) blk_store(out, p.gf, ((size_t)row * 3 + c) * NF, t);
  }
}

int launch_node_fwd(const NodeFwdArgs& a, hipStream_t s) {
  if (a.N <= 0) return 0;
  ScopedTimer t0(TC_LIN, s);
  ScopedTimer t1(TC_LIN1, s);
  node_fwd_kernel<<<cdiv(a.N, 32), 256, 0, s>>>(a);
  LAUNCH_CHECK();
  return 0;
}

int launch_node_bwd(const NodeBwdArgs& a, hipStream_t s) {
  if (a.N <= 0) return 0;
  ScopedTimer t0(TC_LIN, s);
  ScopedTimer t1(TC_LIN1, s);
  node_bwd_kernel<<<cdiv(a.N, 32), 256, 0, s>>>(a);
  LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Small-M form of the fused edge MLP (same contract as mlp128.hip:mlp128_kernel, same MlpArgs): one 4-wave workgroup
// per 32-row tile, wave w = column block w of both stages, weights from L2, hidden tile through LDS.  The persistent
// kernel costs one 512-MFMA chain plus a 132 KiB LDS fill even for a single tile (~35 us); this one runs two 64-MFMA
// chains (~10 us).  Used for the single-molecule / MD-loop regime (MLAseCalculator, ase_interface.py:52-81).
// ---------------------------------------------------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ void mlp_wide_body(const MlpArgs& p, const bool accum, Tile& t) {
  const int M = mlp_rows(p);
  if ((int)blockIdx.x * 32 >= M) return;   // (block-uniform: a tile beyond a device-side row count; never taken otherwise)
  const int row = blockIdx.x * 32 + t.r;
  const int rc = min(row, M - 1);
  const bool live = row < M;

  float4 wf[16];
  load_w(wf, t, p.W1);
  float x[16], hv[16], hin[16];
  blk_load(x, p.X, (size_t)rc * p.ldx, t);
  if (MODE != MODE_FWD) blk_load(hin, p.H, (size_t)rc * p.ldh, t);
  blk_to_tile(x, t);
  __syncthreads();
  acc_to(hv, tile_gemm(t, wf));
  load_w(wf, t, p.W2);
  if (MODE == MODE_FWD) {
    if (p.b1) {
      float b[16];
      blk_load(b, p.b1, 0, t);
#pragma unroll
      for (int k = 0; k < 16; ++k) hv[k] += b[k];
    }
    if (live) blk_store(hv, p.H, (size_t)row * p.ldh, t);
    NN_ACT_BLOCK(hv, 16, p.act);
  } else if (MODE == MODE_TAN2) {
    float t2[16], hd[16];
    blk_load(t2, p.T2, (size_t)rc * p.ldh, t);
    blk_load(hd, p.Hd, (size_t)rc * p.ldh, t);
#pragma unroll
    for (int k = 0; k < 16; ++k)
      hv[k] = fmaf(hv[k], dact_any(hin[k], p.act), t2[k] * d2act_any(hin[k], p.act) * hd[k]);
    if (live) blk_store(hv, p.G, (size_t)row * p.ldh, t);
  } else {
    if (MODE == MODE_TAN && live) blk_store(hv, p.T, (size_t)row * p.ldh, t);
    NN_DACT_MUL_BLOCK(hv, hin, 16, p.act);
  }
  __syncthreads();
  blk_to_tile(hv, t);
  __syncthreads();
  float y[16];
  acc_to(y, tile_gemm(t, wf));
  if (MODE == MODE_FWD && p.b2) {
    float b[16];
    blk_load(b, p.b2, 0, t);
#pragma unroll
    for (int k = 0; k < 16; ++k) y[k] += b[k];
  }
  if (accum) {   // uniform
    float yold[16];
    blk_load(yold, p.Y, (size_t)rc * p.ldy, t);
#pragma unroll
    for (int k = 0; k < 16; ++k) y[k] += yold[k];
  }
  if (live) blk_store(y, p.Y, (size_t)row * p.ldy, t);
}

template <int MODE, bool ACCUM>
__global__ void __launch_bounds__(256) mlp128_wide_kernel(const MlpArgs p) {
  __shared__ __attribute__((aligned(16))) float xs[NODE_LDS_FLOATS];
  Tile t;
  t.xs = xs;
  t.r = threadIdx.x & 31;
  t.h = (threadIdx.x >> 5) & 1;
  t.nb = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  mlp_wide_body<MODE>(p, ACCUM, t);
}

// Two MLPs over the same rows in ONE launch of the small-M form (the single-molecule / MD-loop regime is bound by the number
// of dependent dispatches, ~8 us each).  Forward: phi1 and phi2 are independent -- blockIdx.y picks the MLP.  Adjoint: the
// second term accumulates onto the first one's g_msg rows, which the same lanes of the same workgroup wrote -- run in order.
template <int MODE, bool PAR>
__global__ void __launch_bounds__(256) mlp128_wide_pair_kernel(const MlpPair P) {
  __shared__ __attribute__((aligned(16))) float xs[NODE_LDS_FLOATS];
  Tile t;
  t.xs = xs;
  t.r = threadIdx.x & 31;
  t.h = (threadIdx.x >> 5) & 1;
  t.nb = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (PAR) {   // independent outputs: blockIdx.y picks the MLP
    if (blockIdx.y == 0)
      mlp_wide_body<MODE>(P.a[0], false, t);
    else
      mlp_wide_body<MODE>(P.a[1], false, t);
  } else {     // the second one accumulates onto rows the same lanes of this workgroup just wrote: run in order
    mlp_wide_body<MODE>(P.a[0], false, t);
    __syncthreads();                     // the LDS tile is reused
    mlp_wide_body<MODE>(P.a[1], P.accum[1] != 0, t);
  }
}

// One dense linear Y (+)= X W^T in the same row-local form (no LDS staging of W: a 0.03 GFLOP product is all latency, the
// persistent lin128 kernel spends 17 us on its 64 KiB weight fill alone)
template <bool ACC>
__global__ void __launch_bounds__(256)
lin128_wide_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ W, float* __restrict__ Y, int ldy, int M) {
  __shared__ __attribute__((aligned(16))) float xs[NODE_LDS_FLOATS];
  Tile t;
  t.xs = xs;
  t.r = threadIdx.x & 31;
  t.h = (threadIdx.x >> 5) & 1;
  t.nb = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row = blockIdx.x * 32 + t.r;
  const int rc = min(row, M - 1);
  float4 wf[16];
  load_w(wf, t, W);
  float x[16], y[16];
  blk_load(x, X, (size_t)rc * ldx, t);
  blk_to_tile(x, t);
  __syncthreads();
  if (ACC) blk_load(x, Y, (size_t)rc * ldy, t);
  acc_to(y, tile_gemm(t, wf));
  if (ACC) {
#pragma unroll
    for (int k = 0; k < 16; ++k) y[k] += x[k];
  }
  if (row < M) blk_store(y, Y, (size_t)row * ldy, t);
}
int launch_lin_wide(const float* X, int ldx, const float* W, float* Y, int ldy, int M, bool acc, hipStream_t s) {
  if (M <= 0) return 0;
  ScopedTimer t0(TC_LIN, s);
  if (acc)
    lin128_wide_kernel<true><<<cdiv(M, 32), 256, 0, s>>>(X, ldx, W, Y, ldy, M);
  else
    lin128_wide_kernel<false><<<cdiv(M, 32), 256, 0, s>>>(X, ldx, W, Y, ldy, M);
  LAUNCH_CHECK();
  return 0;
}

static bool wide_split(const MlpArgs& a) {
  return a.W1_img && a.W2_img && a.act == NNHIP_ACT_SILU && split_products_enabled();
}

int launch_mlp_wide(int mode, bool accum, const MlpArgs& a, hipStream_t s) {
  if (wide_split(a)) return launch_mlp_wide_split(mode, accum, a, s);
  const int n_tiles = cdiv(a.M, 32);
  if (mode == MODE_FWD && !accum)
    mlp128_wide_kernel<MODE_FWD, false><<<n_tiles, 256, 0, s>>>(a);
  else if (mode == MODE_BWD && !accum)
    mlp128_wide_kernel<MODE_BWD, false><<<n_tiles, 256, 0, s>>>(a);
  else if (mode == MODE_BWD && accum)
    mlp128_wide_kernel<MODE_BWD, true><<<n_tiles, 256, 0, s>>>(a);
  else if (mode == MODE_TAN && !accum)
    mlp128_wide_kernel<MODE_TAN, false><<<n_tiles, 256, 0, s>>>(a);
  else if (mode == MODE_TAN && accum)
    mlp128_wide_kernel<MODE_TAN, true><<<n_tiles, 256, 0, s>>>(a);
  else if (mode == MODE_TAN2 && !accum)
    mlp128_wide_kernel<MODE_TAN2, false><<<n_tiles, 256, 0, s>>>(a);
  else if (mode == MODE_TAN2 && accum)
    mlp128_wide_kernel<MODE_TAN2, true><<<n_tiles, 256, 0, s>>>(a);
  else
    return NNHIP_E_INVALID;
  LAUNCH_CHECK();
  return 0;
}

int launch_mlp_wide_pair(int mode, const MlpPair& P, hipStream_t s) {
  if (wide_split(P.a[0]) && wide_split(P.a[1])) return launch_mlp_wide_pair_split(mode, P, s);
  const int n_tiles = cdiv(P.a[0].M, 32);
  const bool par = !P.accum[1];   // two independent MLPs: side by side (blockIdx.y)
#define WIDE_PAIR(M_)                                                                  \
  if (mode == M_) {                                                                    \
    if (par)                                                                           \
      mlp128_wide_pair_kernel<M_, true><<<dim3(n_tiles, 2), 256, 0, s>>>(P);           \
    else                                                                               \
      mlp128_wide_pair_kernel<M_, false><<<n_tiles, 256, 0, s>>>(P);                   \
    LAUNCH_CHECK();                                                                    \
    return 0;                                                                          \
  }
  WIDE_PAIR(MODE_FWD)
  WIDE_PAIR(MODE_BWD)
  WIDE_PAIR(MODE_TAN)
  WIDE_PAIR(MODE_TAN2)
#undef WIDE_PAIR
  return NNHIP_E_INVALID;
}

// ---------------------------------------------------------------------------------------------------------------
// direct_force head (newtonnet/models/output.py:115-132, scalers.py:55-56):
//   d = Linear(silu(Linear(silu(Linear(atom_node)))))  [N][F];   force[i][k] = scale[z_i] * < d[i] , force_node[i][k] >
// The first two linears run in the row-local MLP kernel above (biased form), the third in lin128; this is the tail.
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
direct_force_tail_kernel(const float* __restrict__ d, const float* __restrict__ force_node, const float* __restrict__ scale,
                         const int64_t* __restrict__ z, int n_atoms, float* __restrict__ out) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n_atoms) return;
  const int lane = threadIdx.x & 63;
  const float2 dv = ld2(d + (size_t)i * NF + 2 * lane);
  float s[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float2 f = ld2(force_node + ((size_t)i * 3 + k) * NF + 2 * lane);
    s[k] = wave_sum(fmaf(dv.x, f.x, dv.y * f.y));
  }
  if (lane == 0) {
    const float sc = scale ? scale[clamp_species(z[i])] : 1.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) out[(size_t)i * 3 + k] = s[k] * sc;
  }
}

extern "C" int nnhip_direct_force(const float* atom_node, const float* force_node, const int64_t* z, const float* w0,
                                  const float* b0, const float* w2, const float* b2, const float* w4, const float* b4,
                                  const float* scale, int32_t activation, int32_t n_atoms, float* scratch, float* out,
                                  void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (!atom_node || !force_node || !z || !w0 || !b0 || !w2 || !b2 || !w4 || !b4 || !scratch || !out || n_atoms < 0 ||
      activation < NNHIP_ACT_SILU || activation > NNHIP_ACT_SSP) {
    nnhip_set_error("nnhip_direct_force: bad arguments");
    return NNHIP_E_INVALID;
  }
  if (n_atoms == 0) return NNHIP_OK;
  float* e1 = scratch;
  float* e2 = scratch + (size_t)n_atoms * NF;
  float* d = scratch + 2 * (size_t)n_atoms * NF;
  MlpArgs a;
  memset(&a, 0, sizeof(a));
  a.X = atom_node;
  a.W1 = w0;
  a.W2 = w2;
  a.H = e1;
  a.Y = e2;
  a.M = n_atoms;
  a.ldx = a.ldh = a.ldy = NF;
  a.b1 = b0;
  a.b2 = b2;
  a.act = activation;
  int rc = launch_mlp(MODE_FWD, false, a, s);
  if (rc) return rc;
  LinArgs l;
  memset(&l, 0, sizeof(l));
  l.g[0] = {e2, w4, d, b4, nullptr};
  l.M = n_atoms;
  l.lda = l.ldc = NF;
  l.act = activation;
  rc = launch_lin(PRO_SILU, EPI_BIAS, l, 1, s);
  if (rc) return rc;
  direct_force_tail_kernel<<<cdiv(n_atoms, 4), 256, 0, s>>>(d, force_node, scale, z, n_atoms, out);
  LAUNCH_CHECK();
  return NNHIP_OK;
}
