// Fused two-layer 128->128->128 edge MLP on the matrix cores, and its adjoint (gfx950, exact fp32 MFMA).
//
// Replaces equiv_message1 / equiv_message2 = Linear -> SiLU -> Linear (no bias) of InteractionNet
// (newtonnet/models/newtonnet.py:188-197, called at :218,:222) and the matching pair of x W products of the
// reverse sweep, which carry ~87 % of the model's FLOPs at M = n_edges.
//
//   forward  (MODE_FWD):  H = X W1^T (stored: the adjoint needs it);  Y = silu(H) W2^T
//   adjoint  (MODE_BWD):  T = X W1^T;  G = T * silu'(H) (H read back);  Y (+)= G W2^T
//                         with X = g_phi, W1 := V2^T, W2 := V1^T  ->  Y = g_msg
//
// Why a second kernel next to lin128.hip: with separate launches the hidden tile makes a full HBM round trip
// ([E,128] written, then read back) between the two GEMMs; here it never leaves the registers.
// The trick is to compute everything TRANSPOSED: D^T[feature][edge] = W . X^T, i.e. the weights are the MFMA A
// operand (from LDS) and the activations the B operand.  The 32x32 accumulator layout then gives each lane ONE edge
// (column = lane & 31) and 16 of a block's 32 features in its registers (row = (reg&3) + 8 (reg>>2) + 4 (lane>>5)) --
// which is exactly the B-operand layout the second GEMM needs (k-pair {n, n+4} across the two lane halves), so the
// stage-1 accumulators feed stage 2 directly after the in-register activation, like P in a flash-attention PV
// step.  Every global access of a tile is a 16-byte vector per lane (32 contiguous bytes per edge row per
// instruction): 16 loads + 32 stores per 512 MFMAs.
//
// One persistent 8-wave workgroup per CU keeps BOTH weight matrices in LDS (2 x 66 KiB, 16-B row pad -> conflict-free
// ds_read_b128); each wave owns 32-edge tiles; the next tile's X fragment is requested right after stage 1 (its
// registers are dead during stage 2), so the 256 stage-2 MFMAs cover its latency.
#include <stdlib.h>
#include <string.h>

#include "common.h"

#define MW_LD 132
#define MLP_LDS_BYTES (2 * NF * MW_LD * 4)

__device__ __forceinline__ void mlp_load_x(float4 (&x)[16], const float* X, int ldx, int row, int h) {
  const float4* xp = reinterpret_cast<const float4*>(X + (size_t)row * ldx + 4 * h);
#pragma unroll
  for (int t = 0; t < 16; ++t) x[t] = xp[2 * t];  // features 8t + 4h + {0..3}
}

template <int MODE, bool ACCUM>
__global__ void __launch_bounds__(512, 2) mlp128_kernel(const MlpArgs p) {
  extern __shared__ __attribute__((aligned(16))) float wl[];
  float* w1s = wl;
  float* w2s = wl + NF * MW_LD;

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int n_tiles = (p.M + 31) >> 5;
  const float* w1row = w1s + r * MW_LD + 4 * h;
  const float* w2row = w2s + r * MW_LD + 4 * h;

  // Tile -> wave map.  Waves w and w + 4 of a workgroup share a SIMD (waves are dealt to the 4 SIMDs cyclically), and the
  // matrix pipe of a SIMD is what bounds this kernel, so the tiles are dealt to SIMDs first and to the two waves of a
  // SIMD second: when the tile count is not a multiple of the wave count the surplus tiles land on DIFFERENT SIMDs
  // (e.g. 4891 tiles, 1024 SIMDs: 3 + 2 on 795 of them, 2 + 2 on the rest -- not 3 + 3 on 400 and 2 + 2 elsewhere).
  const int n_simd = gridDim.x * 4;
  int tile = (wave >> 2) * n_simd + blockIdx.x * 4 + (wave & 3);
  const int tile_step = gridDim.x * 8;
  float4 x[16];   // first X fragment is requested before the weights are staged: its latency hides under the LDS fill
  mlp_load_x(x, p.X, p.ldx, min((min(tile, n_tiles - 1) << 5) + r, p.M - 1), h);
  for (int idx = threadIdx.x; idx < NF * (NF / 4); idx += 512) {
    const int n = idx >> 5, k4 = idx & 31;
    *reinterpret_cast<float4*>(&w1s[n * MW_LD + k4 * 4]) = reinterpret_cast<const float4*>(p.W1)[idx];
    *reinterpret_cast<float4*>(&w2s[n * MW_LD + k4 * 4]) = reinterpret_cast<const float4*>(p.W2)[idx];
  }
  __syncthreads();
  if (tile >= n_tiles) return;
  for (; tile < n_tiles; tile += tile_step) {
    const int e = (tile << 5) + r;           // this lane's edge (both halves of the wave share it)
    const int ec = min(e, p.M - 1);
    const bool live = e < p.M;
    float hs[4][16];                         // stage-1 result -> activation, feature nb*32 + (k&3) + 8(k>>2) + 4h

    // ---------------- stage 1: H^T = W1 . X^T  (4 blocks of 32 features)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      float4 hin[4];
      if (MODE == MODE_BWD) {  // forward pre-activation of this block, same fragment layout as the stores below
        const float4* hp = reinterpret_cast<const float4*>(p.H + (size_t)ec * p.ldh + nb * 32 + 4 * h);
#pragma unroll
        for (int q = 0; q < 4; ++q) hin[q] = hp[2 * q];
      }
      f32x16 acc;
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[k] = 0.f;
      const float* wr = w1row + nb * 32 * MW_LD;
      float4 a = *reinterpret_cast<const float4*>(wr);
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        float4 an;
        if (t < 15) an = *reinterpret_cast<const float4*>(wr + 8 * (t + 1));
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, x[t].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, x[t].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, x[t].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, x[t].w, acc, 0, 0, 0);
        if (t < 15) a = an;
      }
      __builtin_amdgcn_sched_barrier(0);
      if (MODE == MODE_FWD) {
        if (live) {
          float4* hp = reinterpret_cast<float4*>(p.H + (size_t)e * p.ldh + nb * 32 + 4 * h);
#pragma unroll
          for (int q = 0; q < 4; ++q) hp[2 * q] = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) hs[nb][k] = silu_f(acc[k]);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          hs[nb][4 * q] = acc[4 * q] * dsilu_f(hin[q].x);
          hs[nb][4 * q + 1] = acc[4 * q + 1] * dsilu_f(hin[q].y);
          hs[nb][4 * q + 2] = acc[4 * q + 2] * dsilu_f(hin[q].z);
          hs[nb][4 * q + 3] = acc[4 * q + 3] * dsilu_f(hin[q].w);
        }
      }
    }

    // X of the NEXT tile: its registers are dead now; stage 2 (256 MFMAs) hides the latency.  (asm volatile pins the
    // request here, ahead of this tile's stage-2 stores in the in-order vmcnt queue.)
    {
      const int nrow = min((min(tile + tile_step, n_tiles - 1) << 5) + r, p.M - 1);
      mlp_load_x(x, p.X, p.ldx, nrow, h);
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---------------- stage 2: Y^T = W2 . act^T ; the B operand is the stage-1 register tile
#pragma unroll
    for (int nb2 = 0; nb2 < 4; ++nb2) {
      float4 yold[4];
      if (ACCUM) {
        const float4* yp = reinterpret_cast<const float4*>(p.Y + (size_t)ec * p.ldy + nb2 * 32 + 4 * h);
#pragma unroll
        for (int q = 0; q < 4; ++q) yold[q] = yp[2 * q];
      }
      f32x16 acc;
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[k] = 0.f;
      const float* wr = w2row + nb2 * 32 * MW_LD;
      float4 a = *reinterpret_cast<const float4*>(wr);
#pragma unroll
      for (int T = 0; T < 16; ++T) {  // k-group T: features (T>>2)*32 + 8 (T&3) + 4h + {0..3} = hs[T>>2][4 (T&3) + c]
        float4 an;
        if (T < 15) an = *reinterpret_cast<const float4*>(wr + 8 * (T + 1));
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, hs[T >> 2][4 * (T & 3) + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, hs[T >> 2][4 * (T & 3) + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, hs[T >> 2][4 * (T & 3) + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, hs[T >> 2][4 * (T & 3) + 3], acc, 0, 0, 0);
        if (T < 15) a = an;
      }
      __builtin_amdgcn_sched_barrier(0);
      if (live) {
        float4* yp = reinterpret_cast<float4*>(p.Y + (size_t)e * p.ldy + nb2 * 32 + 4 * h);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float4 v = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
          if (ACCUM) {
            v.x += yold[q].x;
            v.y += yold[q].y;
            v.z += yold[q].z;
            v.w += yold[q].w;
          }
          yp[2 * q] = v;
        }
      }
    }
  }
}

template <int MODE, bool ACCUM>
static int launch_mlp_t(const MlpArgs& a, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    HIP_TRY(hipFuncSetAttribute((const void*)mlp128_kernel<MODE, ACCUM>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                MLP_LDS_BYTES));
    attr_set = true;
  }
  const int n_tiles = (a.M + 31) / 32;
  int blocks = cdiv(n_tiles, 8);
  if (blocks > 256) blocks = 256;  // one persistent workgroup per CU (135 KiB of LDS each)
  mlp128_kernel<MODE, ACCUM><<<blocks, 512, MLP_LDS_BYTES, s>>>(a);
  LAUNCH_CHECK();
  return 0;
}

int launch_mlp_wide(int mode, bool accum, const MlpArgs& a, hipStream_t s);   // node128.hip

#define MLP_WIDE_MAX_TILES 1536   // up to ~49k rows one workgroup per tile beats the persistent form (tools/bench_mlp.py)

int launch_mlp(int mode, bool accum, const MlpArgs& a, hipStream_t s) {
  if (a.M <= 0) return 0;
  ScopedTimer t0(TC_LIN, s);
  ScopedTimer t1((a.b1 || a.b2) ? TC_LIN1 : TC_MLP, s);   // biased form = node MLP / energy head (node-level class)
  static const int wide_max = getenv("NNHIP_MLP_WIDE_TILES") ? atoi(getenv("NNHIP_MLP_WIDE_TILES")) : MLP_WIDE_MAX_TILES;
  if (cdiv(a.M, 32) <= wide_max || a.b1 || a.b2) return launch_mlp_wide(mode, accum, a, s);
  if (mode == MODE_FWD && !accum) return launch_mlp_t<MODE_FWD, false>(a, s);
  if (mode == MODE_BWD && !accum) return launch_mlp_t<MODE_BWD, false>(a, s);
  if (mode == MODE_BWD && accum) return launch_mlp_t<MODE_BWD, true>(a, s);
  nnhip_set_error("launch_mlp: unsupported mode %d/%d", mode, (int)accum);
  return NNHIP_E_INVALID;
}

// C ABI (include/newtonnet_hip.h)
extern "C" int nnhip_mlp128(const float* X, int32_t ldx, const float* W1, const float* W2, float* H, int32_t ldh, float* Y,
                            int32_t ldy, int32_t M, int32_t mode, int32_t accumulate, void* stream) {
  if (!X || !W1 || !W2 || !H || !Y || M < 0 || ldx < NF || ldh < NF || ldy < NF || (ldx & 3) || (ldh & 3) || (ldy & 3)) {
    nnhip_set_error("nnhip_mlp128: bad arguments");
    return NNHIP_E_INVALID;
  }
  MlpArgs a;
  a.X = X;
  a.W1 = W1;
  a.W2 = W2;
  a.H = H;
  a.Y = Y;
  a.M = M;
  a.ldx = ldx;
  a.ldh = ldh;
  a.ldy = ldy;
  a.b1 = a.b2 = nullptr;
  return launch_mlp(mode, accumulate != 0, a, (hipStream_t)stream);
}
