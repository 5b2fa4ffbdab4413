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

#include <atomic>

#include "common.h"

#ifndef MLP_NT_H
#define MLP_NT_H 1   // hidden pre-activations bypass the caches (streaming): they would only evict phi / msg
#endif
#ifndef MLP_NT_HL
#define MLP_NT_HL 1   // ... and read back by the adjoint with the same hint
#endif
#define MW_LD 132
#define MLP_LDS_BYTES (2 * NF * MW_LD * 4)
#ifndef MLP_WAVES
#define MLP_WAVES 8   // waves per persistent workgroup (tooling: -DMLP_WAVES=4 = one wave per SIMD, half the register file free)
#endif
#define MLP_THREADS (64 * MLP_WAVES)

#ifndef MLP_NT_X
#define MLP_NT_X 0   // streaming loads of stage-1 inputs on their last read: measured 5 % SLOWER (0.935 -> 0.984 ms), off
#endif
template <bool NT = false>
__device__ __forceinline__ void mlp_load_x(float4 (&x)[16], const float* X, int ldx, int row, int h) {
  const float4* xp = reinterpret_cast<const float4*>(X + (size_t)row * ldx + 4 * h);
#pragma unroll
  for (int t = 0; t < 16; ++t)   // features 8t + 4h + {0..3}
    x[t] = NT ? ld4_nt(reinterpret_cast<const float*>(xp + 2 * t)) : xp[2 * t];
}

// Stage both weight matrices of one MLP: all 16 requests of a thread are in flight before the first LDS write.
// (Macros, not functions: the fragments must stay in registers.)
#if MLP_WAVES == 8
#define MLP_W_LD(q, W1p, W2p)                                                                   \
  const float4 w1v##q = reinterpret_cast<const float4*>(W1p)[threadIdx.x + 512 * q];             \
  const float4 w2v##q = reinterpret_cast<const float4*>(W2p)[threadIdx.x + 512 * q];
#define MLP_W_ST(q)                                                                              \
  *reinterpret_cast<float4*>(&w1s[((threadIdx.x + 512 * q) >> 5) * MW_LD + (threadIdx.x & 31) * 4]) = w1v##q; \
  *reinterpret_cast<float4*>(&w2s[((threadIdx.x + 512 * q) >> 5) * MW_LD + (threadIdx.x & 31) * 4]) = w2v##q;
#define MLP_W_REQUEST(W1p, W2p)                                                                  \
  MLP_W_LD(0, W1p, W2p) MLP_W_LD(1, W1p, W2p) MLP_W_LD(2, W1p, W2p) MLP_W_LD(3, W1p, W2p)        \
  MLP_W_LD(4, W1p, W2p) MLP_W_LD(5, W1p, W2p) MLP_W_LD(6, W1p, W2p) MLP_W_LD(7, W1p, W2p)        \
  __builtin_amdgcn_sched_barrier(0);
#define MLP_W_COMMIT() MLP_W_ST(0) MLP_W_ST(1) MLP_W_ST(2) MLP_W_ST(3) MLP_W_ST(4) MLP_W_ST(5) MLP_W_ST(6) MLP_W_ST(7)
#else   // tooling variant: plain staging loop
#define MLP_W_REQUEST(W1p, W2p)                                                                  \
  for (int q_ = threadIdx.x; q_ < NF * NF / 4; q_ += MLP_THREADS) {                              \
    *reinterpret_cast<float4*>(&w1s[(q_ >> 5) * MW_LD + (q_ & 31) * 4]) = reinterpret_cast<const float4*>(W1p)[q_]; \
    *reinterpret_cast<float4*>(&w2s[(q_ >> 5) * MW_LD + (q_ & 31) * 4]) = reinterpret_cast<const float4*>(W2p)[q_]; \
  }
#define MLP_W_COMMIT()
#endif

// One launch runs P.n (1 or 2) MLPs of the same row count back to back ("phases": phi1 then phi2, or the two adjoint
// terms of g_msg).  A launch costs ~15 us before the matrix pipes are busy (every wave's first X tile -- 33 MB -- and
// 256 copies of the weights are requested at once) and the same again in stragglers at the end; the second phase only
// restages the weights (L2 hits, the next X tile is already in flight) and reuses the running pipeline.
// ACCUM_LAST: the last phase adds its result to Y (P.accum is checked by the host).  GEN_ACT: an activation other than SiLU
// (P.a[0].act); the SiLU instantiation is untouched by it.
template <int MODE, bool ACCUM_LAST, bool GEN_ACT>
__global__ void __launch_bounds__(MLP_THREADS, MLP_WAVES / 4) mlp128_kernel(const MlpPair P) {
  extern __shared__ __attribute__((aligned(16))) float wl[];
  float* w1s = wl;
  float* w2s = wl + NF * MW_LD;

#ifdef MLP_CLOCK_DEBUG
  const long long dbg_wstart = wall_clock64();
#endif
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int M = mlp_rows(P.a[0]);
  if (M <= 0) return;   // (uniform; only possible with a device-side count)
  const int n_tiles = (M + 31) >> 5;
  const float* w1row = w1s + r * MW_LD + 4 * h;
  const float* w2row = w2s + r * MW_LD + 4 * h;

  // Tile -> wave map.  Waves w and w + 4 of a workgroup share a SIMD (waves are dealt to the 4 SIMDs cyclically), and the
  // matrix pipe of a SIMD is what bounds this kernel, so the tiles are dealt to SIMDs first and to the two waves of a
  // SIMD second: when the tile count is not a multiple of the wave count the surplus tiles land on DIFFERENT SIMDs
  // (e.g. 4891 tiles, 1024 SIMDs: 3 + 2 on 795 of them, 2 + 2 on the rest -- not 3 + 3 on 400 and 2 + 2 elsewhere).
  const int n_simd = gridDim.x * 4;
  const int tile0 = (wave >> 2) * n_simd + blockIdx.x * 4 + (wave & 3);
  const int tile_step = gridDim.x * MLP_WAVES;
  float4 x[16];
  {
    // weights first (L2 hits, returned first by the in-order vmcnt queue), then this wave's first X fragment: the LDS fill
    // and the barrier only wait for the weights, and stage 1 starts consuming X as its 16 pieces arrive
    MLP_W_REQUEST(P.a[0].W1, P.a[0].W2)
    if (MLP_NT_X && (MODE != MODE_FWD || P.n == 1))
      mlp_load_x<true>(x, P.a[0].X, P.a[0].ldx, min((min(tile0, n_tiles - 1) << 5) + r, M - 1), h);
    else
      mlp_load_x<false>(x, P.a[0].X, P.a[0].ldx, min((min(tile0, n_tiles - 1) << 5) + r, M - 1), h);
    __builtin_amdgcn_sched_barrier(0);
    MLP_W_COMMIT()
  }
  __syncthreads();
#ifdef MLP_CLOCK_DEBUG   // tooling: shader clock vs the 100 MHz wall clock over the tile loop of one wave
  const long long dbg_c0 = clock64(), dbg_w0 = wall_clock64();
  int dbg_tiles = 0;
#endif
  for (int ph = 0; ph < P.n; ++ph) {
    // per-phase arguments picked with selects (indexing the kernel-argument array with `ph` would spill it to scratch)
    struct { const float* X; float* H; float* Y; int ldx, ldh, ldy; } p;
    const bool h_frag = P.a[0].h_frag != 0;   // (both phases of a launch use the same H layout)
    p.X = ph ? P.a[1].X : P.a[0].X;
    p.H = ph ? P.a[1].H : P.a[0].H;
    p.Y = ph ? P.a[1].Y : P.a[0].Y;
    p.ldx = ph ? P.a[1].ldx : P.a[0].ldx;
    p.ldh = ph ? P.a[1].ldh : P.a[0].ldh;
    p.ldy = ph ? P.a[1].ldy : P.a[0].ldy;
    const bool accum = ACCUM_LAST && ph == P.n - 1;            // uniform
    const bool more = ph + 1 < P.n;                            // (P.n <= 2: a following phase is always a[1])
    if (ph > 0) {
      __syncthreads();                     // every wave is done with the previous phase's weights
      MLP_W_REQUEST(P.a[1].W1, P.a[1].W2)
      MLP_W_COMMIT()
      __syncthreads();
    }
  for (int tile = tile0; tile < n_tiles; tile += tile_step) {
#ifdef MLP_CLOCK_DEBUG
    ++dbg_tiles;
#endif
    const int e = (tile << 5) + r;           // this lane's edge (both halves of the wave share it)
    const int ec = min(e, M - 1);
    const bool live = e < M;
    float hs[4][16];                         // stage-1 result -> activation, feature nb*32 + (k&3) + 8(k>>2) + 4h

    // ---------------- stage 1: H^T = W1 . X^T  (4 blocks of 32 features)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      float4 hin[4];
      if (MODE != MODE_FWD) {  // forward pre-activation of this block, same fragment layout as the stores below
        const float4* hp = h_frag ? reinterpret_cast<const float4*>(p.H) + ((size_t)tile * 4 + nb) * 256 + lane
                                  : reinterpret_cast<const float4*>(p.H + (size_t)ec * p.ldh + nb * 32 + 4 * h);
        const int hs4 = h_frag ? 64 : 2;
#pragma unroll
        for (int q = 0; q < 4; ++q) hin[q] = MLP_NT_HL ? ld4_nt(reinterpret_cast<const float*>(hp + hs4 * q)) : hp[hs4 * q];
      }
      f32x16 acc;
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[k] = 0.f;
      const float* wr = w1row + nb * 32 * MW_LD;
      // A fragments come from LDS two k-groups ahead of their use; the sched_group_barrier pairs pin the order
      // "1 ds_read, 4 MFMA" (left alone, the compiler sinks each read to just before its first use).
      float4 a0 = *reinterpret_cast<const float4*>(wr), a1 = *reinterpret_cast<const float4*>(wr + 8);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        float4 a2;
        if (t < 14) a2 = *reinterpret_cast<const float4*>(wr + 8 * (t + 2));
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, x[t].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, x[t].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, x[t].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, x[t].w, acc, 0, 0, 0);
        a0 = a1;
        if (t < 14) a1 = a2;
        if (t < 14) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (MODE == MODE_FWD) {
#ifdef ABL_NO_STORE   // tooling only
        if (live && M < 0) {
#else
        if (live || h_frag) {   // (fragment order: rows past M land in the pad32 tail of the region)
#endif
          float4* hp = h_frag ? reinterpret_cast<float4*>(p.H) + ((size_t)tile * 4 + nb) * 256 + lane
                              : reinterpret_cast<float4*>(p.H + (size_t)e * p.ldh + nb * 32 + 4 * h);
          const int hs4 = h_frag ? 64 : 2;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 hv = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
            if (MLP_NT_H)
              st4_nt(reinterpret_cast<float*>(hp + hs4 * q), hv);   // written once, read once by the adjoint much later
            else
              hp[hs4 * q] = hv;
          }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) hs[nb][k] = GEN_ACT ? act_f(acc[k], P.a[0].act) : silu_f(acc[k]);
      } else if (MODE == MODE_TAN2) {
        // tangent of the adjoint: G = dT act'(H) + T2 act''(H) Hd, kept (row-major) for the weight-gradient products
        const float* t2p = (ph ? P.a[1].T2 : P.a[0].T2) + (size_t)ec * p.ldh + nb * 32 + 4 * h;
        const float* hdp = (ph ? P.a[1].Hd : P.a[0].Hd) + (size_t)ec * p.ldh + nb * 32 + 4 * h;
        float* gp = (ph ? P.a[1].G : P.a[0].G) + (size_t)e * p.ldh + nb * 32 + 4 * h;
        const int act = P.a[0].act;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 t2 = ld4(t2p + 8 * q), hd = ld4(hdp + 8 * q);
          float4 gv;
#define MLP_D1(v) (GEN_ACT ? dact_f(v, act) : dsilu_f(v))
#define MLP_D2(v) (GEN_ACT ? d2act_f(v, act) : d2silu_f(v))
          gv.x = fmaf(acc[4 * q], MLP_D1(hin[q].x), t2.x * MLP_D2(hin[q].x) * hd.x);
          gv.y = fmaf(acc[4 * q + 1], MLP_D1(hin[q].y), t2.y * MLP_D2(hin[q].y) * hd.y);
          gv.z = fmaf(acc[4 * q + 2], MLP_D1(hin[q].z), t2.z * MLP_D2(hin[q].z) * hd.z);
          gv.w = fmaf(acc[4 * q + 3], MLP_D1(hin[q].w), t2.w * MLP_D2(hin[q].w) * hd.w);
#undef MLP_D1
#undef MLP_D2
          hs[nb][4 * q] = gv.x;
          hs[nb][4 * q + 1] = gv.y;
          hs[nb][4 * q + 2] = gv.z;
          hs[nb][4 * q + 3] = gv.w;
          if (live) st4(gp + 8 * q, gv);
        }
      } else {
        if (MODE == MODE_TAN && live) {   // keep the stage-1 product (row-major): the tangent sweeps and weight gradients read it
          float* tp = (ph ? P.a[1].T : P.a[0].T) + (size_t)e * p.ldh + nb * 32 + 4 * h;
#pragma unroll
          for (int q = 0; q < 4; ++q) st4(tp + 8 * q, make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          hs[nb][4 * q] = acc[4 * q] * (GEN_ACT ? dact_f(hin[q].x, P.a[0].act) : dsilu_f(hin[q].x));
          hs[nb][4 * q + 1] = acc[4 * q + 1] * (GEN_ACT ? dact_f(hin[q].y, P.a[0].act) : dsilu_f(hin[q].y));
          hs[nb][4 * q + 2] = acc[4 * q + 2] * (GEN_ACT ? dact_f(hin[q].z, P.a[0].act) : dsilu_f(hin[q].z));
          hs[nb][4 * q + 3] = acc[4 * q + 3] * (GEN_ACT ? dact_f(hin[q].w, P.a[0].act) : dsilu_f(hin[q].w));
        }
      }
    }

    // X of the NEXT tile (of this phase, or the first tile of the next phase): its registers are dead now; stage 2
    // (256 MFMAs) hides the latency.  (The sched_barrier pins the request here, ahead of this tile's stage-2 stores in
    // the in-order vmcnt queue.)
    {
      const bool last = tile + tile_step >= n_tiles;
      const float* xn = (last && more) ? P.a[1].X : p.X;
      const int ldn = (last && more) ? P.a[1].ldx : p.ldx;
      const int nt = last ? tile0 : tile + tile_step;
      // last read of this input?  adjoint: always (g_phi is consumed here); forward: msg is read by both phases
      const bool last_use = MODE != MODE_FWD || P.n == 1 || ph == 1 || (last && more);
      if (MLP_NT_X && last_use)
        mlp_load_x<true>(x, xn, ldn, min((min(nt, n_tiles - 1) << 5) + r, M - 1), h);
      else
        mlp_load_x<false>(x, xn, ldn, min((min(nt, n_tiles - 1) << 5) + r, M - 1), h);
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---------------- stage 2: Y^T = W2 . act^T ; the B operand is the stage-1 register tile
#pragma unroll
    for (int nb2 = 0; nb2 < 4; ++nb2) {
      float4 yold[4];
      if (ACCUM_LAST) {
        const float4* yp = reinterpret_cast<const float4*>(p.Y + (size_t)ec * p.ldy + nb2 * 32 + 4 * h);
#pragma unroll
        for (int q = 0; q < 4; ++q) yold[q] = accum ? yp[2 * q] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      f32x16 acc;
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[k] = 0.f;
      const float* wr = w2row + nb2 * 32 * MW_LD;
      float4 a0 = *reinterpret_cast<const float4*>(wr), a1 = *reinterpret_cast<const float4*>(wr + 8);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int T = 0; T < 16; ++T) {  // k-group T: features (T>>2)*32 + 8 (T&3) + 4h + {0..3} = hs[T>>2][4 (T&3) + c]
        float4 a2;
        if (T < 14) a2 = *reinterpret_cast<const float4*>(wr + 8 * (T + 2));
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, hs[T >> 2][4 * (T & 3) + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, hs[T >> 2][4 * (T & 3) + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, hs[T >> 2][4 * (T & 3) + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, hs[T >> 2][4 * (T & 3) + 3], acc, 0, 0, 0);
        a0 = a1;
        if (T < 14) a1 = a2;
        if (T < 14) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#ifdef ABL_NO_STORE
      if (live && M < 0) {
#else
      if (live) {
#endif
        float4* yp = reinterpret_cast<float4*>(p.Y + (size_t)e * p.ldy + nb2 * 32 + 4 * h);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float4 v = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
          if (ACCUM_LAST) {
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
#ifdef MLP_CLOCK_DEBUG
  if ((blockIdx.x == 0 || blockIdx.x == 100 || blockIdx.x == 255) && (threadIdx.x & 63) == 0 && (wave == 0 || wave == 7)) {
    const long long wend = wall_clock64();
    const long long dc = clock64() - dbg_c0, dw = wend - dbg_w0;
    printf("mlp128 mode %d blk %3d w%d: start %lld  staged +%.2f us  loop %.2f us (%d tiles, %.3f GHz)  end %lld\n", MODE,
           (int)blockIdx.x, wave, dbg_wstart, (dbg_w0 - dbg_wstart) / 100.0, dw / 100.0, dbg_tiles, dc / (dw * 10.0), wend);
  }
#endif
}

template <int MODE, bool ACCUM_LAST, bool GEN_ACT>
static int launch_mlp_t(const MlpPair& a, hipStream_t s) {
  // one-time kernel attribute; a function-local static is initialised exactly once even with concurrent host threads
  static const hipError_t attr_rc = hipFuncSetAttribute((const void*)mlp128_kernel<MODE, ACCUM_LAST, GEN_ACT>,
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, MLP_LDS_BYTES);
  HIP_TRY(attr_rc);
  const int n_tiles = (a.a[0].M + 31) / 32;
  int blocks = cdiv(n_tiles, MLP_WAVES);
  if (blocks > 256) blocks = 256;  // one persistent workgroup per CU (135 KiB of LDS each)
  mlp128_kernel<MODE, ACCUM_LAST, GEN_ACT><<<blocks, MLP_THREADS, MLP_LDS_BYTES, s>>>(a);
  LAUNCH_CHECK();
  return 0;
}

int launch_mlp_wide(int mode, bool accum, const MlpArgs& a, hipStream_t s);   // node128.hip
int launch_mlp_wide_pair(int mode, const MlpPair& P, hipStream_t s);

#define MLP_WIDE_MAX_TILES 1536   // up to ~49k rows one workgroup per tile beats the persistent form (tools/bench_mlp.py)
#define MLPS_WIDE_MAX_TILES 832   // ... ~27k rows with split-f16 products (config-2 recipe: 611 tiles 0.189 vs 0.220 ms per step
                                  // row-local vs persistent, 1223 tiles 0.309 vs 0.242)

int launch_mlp_split(int mode, bool accum_last, const MlpPair& P, hipStream_t s);   // mlp128s.hip

// The forward / adjoint launches of the persistent kernel take the split-f16 product form (mlp128s.hip: 5x less matrix-pipe
// time, and closer to fp64 than the fp32 MFMA chain below); NNHIP_MLP_SPLIT=0 keeps them on v_mfma_f32_32x32x2_f32 (tooling,
// A/B).  The other activations of the factory run the fp32 form.
extern "C" int nnhip_split_products(void) { return split_products_enabled() ? 1 : 0; }
bool split_products_enabled() {
  static const bool on = !(getenv("NNHIP_MLP_SPLIT") && atoi(getenv("NNHIP_MLP_SPLIT")) == 0);
  return on;
}

static int launch_mlp_dispatch(int mode, bool accum_last, const MlpPair& P, hipStream_t s) {
  if (split_products_enabled() && P.a[0].act == NNHIP_ACT_SILU) return launch_mlp_split(mode, accum_last, P, s);
  if (P.a[0].act != NNHIP_ACT_SILU) {
    if (mode == MODE_FWD && !accum_last) return launch_mlp_t<MODE_FWD, false, true>(P, s);
    if (mode == MODE_BWD && !accum_last) return launch_mlp_t<MODE_BWD, false, true>(P, s);
    if (mode == MODE_BWD && accum_last) return launch_mlp_t<MODE_BWD, true, true>(P, s);
    if (mode == MODE_TAN && !accum_last) return launch_mlp_t<MODE_TAN, false, true>(P, s);
    if (mode == MODE_TAN && accum_last) return launch_mlp_t<MODE_TAN, true, true>(P, s);
    if (mode == MODE_TAN2 && !accum_last) return launch_mlp_t<MODE_TAN2, false, true>(P, s);
    if (mode == MODE_TAN2 && accum_last) return launch_mlp_t<MODE_TAN2, true, true>(P, s);
  }
  if (mode == MODE_FWD && !accum_last) return launch_mlp_t<MODE_FWD, false, false>(P, s);
  if (mode == MODE_BWD && !accum_last) return launch_mlp_t<MODE_BWD, false, false>(P, s);
  if (mode == MODE_BWD && accum_last) return launch_mlp_t<MODE_BWD, true, false>(P, s);
  if (mode == MODE_TAN && !accum_last) return launch_mlp_t<MODE_TAN, false, false>(P, s);
  if (mode == MODE_TAN && accum_last) return launch_mlp_t<MODE_TAN, true, false>(P, s);
  if (mode == MODE_TAN2 && !accum_last) return launch_mlp_t<MODE_TAN2, false, false>(P, s);
  if (mode == MODE_TAN2 && accum_last) return launch_mlp_t<MODE_TAN2, true, false>(P, s);
  nnhip_set_error("launch_mlp: unsupported mode %d/%d", mode, (int)accum_last);
  return NNHIP_E_INVALID;
}

int mlp_wide_max_tiles_silu() {   // (nnhip_config: below this many 32-row tiles the edge MLPs take the row-local form)
  static const int env_max = getenv("NNHIP_MLP_WIDE_TILES") ? atoi(getenv("NNHIP_MLP_WIDE_TILES")) : -1;   // (read once: no getenv on a call path)
  return env_max >= 0 ? env_max : (split_products_enabled() ? MLPS_WIDE_MAX_TILES : MLP_WIDE_MAX_TILES);
}
static bool mlp_use_wide(const MlpArgs& a) {
  static const int env_max = getenv("NNHIP_MLP_WIDE_TILES") ? atoi(getenv("NNHIP_MLP_WIDE_TILES")) : -1;
  const int wide_max = env_max >= 0 ? env_max
                                    : (split_products_enabled() && a.act == NNHIP_ACT_SILU ? MLPS_WIDE_MAX_TILES : MLP_WIDE_MAX_TILES);
  return cdiv(a.M, 32) <= wide_max || a.b1 || a.b2;
}

int launch_mlp(int mode, bool accum, const MlpArgs& a, hipStream_t s) {
  if (a.M <= 0) return 0;
  ScopedTimer t0(TC_LIN, s);
  ScopedTimer t1((a.b1 || a.b2) ? TC_LIN1 : TC_MLP, s);   // biased form = node MLP / energy head (node-level class)
  if (mlp_use_wide(a)) {
    MlpArgs w = a;
    if (w.h_frag) w.ldh = NF;   // the row-local kernel keeps H row-major inside the same pad32(M) x 128 region
    return launch_mlp_wide(mode, accum, w, s);
  }
  if (mode < MODE_FWD || mode > MODE_TAN2 || ((mode == MODE_TAN || mode == MODE_TAN2) && a.h_frag)) {
    nnhip_set_error("launch_mlp: unsupported mode %d", mode);
    return NNHIP_E_INVALID;
  }
  MlpPair P;
  P.a[0] = P.a[1] = a;
  P.n = 1;
  P.accum[0] = P.accum[1] = accum ? 1 : 0;
  if (split_products_enabled() && mlp_regw_serves(mode, P)) {   // the single-MLP adjoint with register-resident weights (mlp128r.hip)
    ScopedTimer t2(TC_MLP_ONEPASS, s);
    return launch_mlp_regw(mode, P, s);
  }
  return launch_mlp_dispatch(mode, accum, P, s);
}

// Two MLPs over the same rows in one persistent launch (phi1 | phi2 forward; the two terms of g_msg in the adjoint).
int launch_mlp_pair(int mode, const MlpArgs& a0, bool accum0, const MlpArgs& a1, bool accum1, hipStream_t s) {
  if (a0.M != a1.M || mode < MODE_FWD || mode > MODE_TAN2 || accum0 || a0.h_frag != a1.h_frag || a0.act != a1.act ||
      a0.ldh != a1.ldh) {   // (only the last phase may accumulate)
    nnhip_set_error("launch_mlp_pair: bad arguments");
    return NNHIP_E_INVALID;
  }
  if (a0.M <= 0) return 0;
  if (mlp_use_wide(a0) || mlp_use_wide(a1)) {
    if (a0.b1 || a0.b2 || a1.b1 || a1.b2 || (mode == MODE_FWD && accum1)) {   // (not a case the path produces)
      const int rc = launch_mlp(mode, accum0, a0, s);
      return rc ? rc : launch_mlp(mode, accum1, a1, s);
    }
    ScopedTimer t0(TC_LIN, s);
    ScopedTimer t1(TC_MLP, s);
    MlpPair W;
    W.a[0] = a0;
    W.a[1] = a1;
    if (W.a[0].h_frag) W.a[0].ldh = W.a[1].ldh = NF;   // row-major inside the same pad32(M) x 128 region
    W.n = 2;
    W.accum[0] = 0;
    W.accum[1] = accum1 ? 1 : 0;
    return launch_mlp_wide_pair(mode, W, s);
  }
  ScopedTimer t0(TC_LIN, s);
  ScopedTimer t1(TC_MLP, s);
  MlpPair P;
  P.a[0] = a0;
  P.a[1] = a1;
  P.n = 2;
  P.accum[0] = accum0 ? 1 : 0;
  P.accum[1] = accum1 ? 1 : 0;
  if (split_products_enabled() && mlp_regw_serves(mode, P)) {   // both MLPs in one pass (mlp128r.hip); timed inside 'mlp128' too
    ScopedTimer t2(TC_MLP_ONEPASS, s);
    return launch_mlp_regw(mode, P, s);
  }
  return launch_mlp_dispatch(mode, accum1, P, s);
}

// C ABI (include/newtonnet_hip.h)
extern "C" int nnhip_mlp128(const float* X, int32_t ldx, const float* W1, const float* W2, float* H, int32_t ldh, float* Y,
                            int32_t ldy, int32_t M, int32_t mode, int32_t accumulate, void* stream) {
  if (!X || !W1 || !W2 || !H || !Y || M < 0 || ldx < NF || ldh < NF || ldy < NF || (ldx & 3) || (ldh & 3) || (ldy & 3)) {
    nnhip_set_error("nnhip_mlp128: bad arguments");
    return NNHIP_E_INVALID;
  }
  MlpArgs a;
  memset(&a, 0, sizeof(a));
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
  a.act = NNHIP_ACT_SILU;
  a.h_frag = 0;   // the C ABI exposes H row-major
  return launch_mlp(mode, accumulate != 0, a, (hipStream_t)stream);
}

// Extended form: biases, any activation of the factory, and the training modes (include/newtonnet_hip.h: nnhip_mlp_desc)
static int desc_to_args(const nnhip_mlp_desc* d, MlpArgs& a, const char* who) {
  if (!d || !d->X || !d->W1 || !d->W2 || !d->H || !d->Y || d->M < 0 || d->ldx < NF || d->ldh < NF || d->ldy < NF ||
      (d->ldx & 3) || (d->ldh & 3) || (d->ldy & 3) || d->mode < MODE_FWD || d->mode > MODE_TAN2 ||
      d->activation < NNHIP_ACT_SILU || d->activation > NNHIP_ACT_SSP || (d->mode == MODE_TAN && !d->T) ||
      (d->mode == MODE_TAN2 && (!d->T2 || !d->Hd || !d->G)) || (d->mode != MODE_FWD && (d->b1 || d->b2))) {
    nnhip_set_error("%s: bad arguments", who);
    return NNHIP_E_INVALID;
  }
  memset(&a, 0, sizeof(a));
  a.X = d->X;
  a.W1 = d->W1;
  a.W2 = d->W2;
  a.H = d->H;
  a.Y = d->Y;
  a.M = d->M;
  a.ldx = d->ldx;
  a.ldh = d->ldh;
  a.ldy = d->ldy;
  a.b1 = d->b1;
  a.b2 = d->b2;
  a.act = d->activation;
  a.T = d->T;
  a.T2 = d->T2;
  a.Hd = d->Hd;
  a.G = d->G;
  if ((d->W1_image == nullptr) != (d->W2_image == nullptr)) {
    nnhip_set_error("%s: W1_image / W2_image come together", who);
    return NNHIP_E_INVALID;
  }
  a.W1_img = (const char*)d->W1_image;
  a.W2_img = (const char*)d->W2_image;
  if (d->precision != 0 && d->precision != 1) {
    nnhip_set_error("%s: precision %d (0 = fp32-grade products, 1 = bf16 operands)", who, d->precision);
    return NNHIP_E_INVALID;
  }
  if (d->precision == 1 && (!split_products_enabled() || d->activation != NNHIP_ACT_SILU || !d->W1_image || d->b1 || d->b2)) {
    nnhip_set_error("%s: the bf16 compute mode serves bias-free SiLU MLPs with bf16 weight images (nnhip_weight_images_bf16)", who);
    return NNHIP_E_UNSUPPORTED;
  }
  a.bf16 = d->precision;
  return NNHIP_OK;
}
// launches that took the bf16 compute mode since the library was loaded (tests assert the form that ran)
static std::atomic<long> g_bf16_mlp_launches{0};
extern "C" int64_t nnhip_bf16_mlp_launches(void) { return (int64_t)g_bf16_mlp_launches.load(); }
extern "C" int nnhip_mlp128_ex(const nnhip_mlp_desc* d, void* stream) {
  MlpArgs a;
  const int rc = desc_to_args(d, a, "nnhip_mlp128_ex");
  if (rc) return rc;
  if (a.bf16 && a.M > 0) g_bf16_mlp_launches.fetch_add(1);
  return launch_mlp(d->mode, d->accumulate != 0, a, (hipStream_t)stream);
}
// Two MLPs over the same M rows in ONE launch (equiv_message1 | equiv_message2 and their adjoints / tangents): same mode and
// activation; only the second may accumulate (then it runs after the first, on the same rows).
extern "C" int nnhip_mlp128_pair_ex(const nnhip_mlp_desc* d0, const nnhip_mlp_desc* d1, void* stream) {
  MlpArgs a0, a1;
  int rc = desc_to_args(d0, a0, "nnhip_mlp128_pair_ex");
  if (rc) return rc;
  rc = desc_to_args(d1, a1, "nnhip_mlp128_pair_ex");
  if (rc) return rc;
  if (d0->mode != d1->mode || d0->M != d1->M || d0->accumulate || d0->activation != d1->activation) {
    nnhip_set_error("nnhip_mlp128_pair_ex: the two MLPs must share mode, M and activation; only the second may accumulate");
    return NNHIP_E_INVALID;
  }
  if (a0.bf16 != a1.bf16) {
    nnhip_set_error("nnhip_mlp128_pair_ex: the two MLPs must share their precision");
    return NNHIP_E_INVALID;
  }
  if (a0.bf16 && a0.M > 0) g_bf16_mlp_launches.fetch_add(1);
  return launch_mlp_pair(d0->mode, a0, false, a1, d1->accumulate != 0, (hipStream_t)stream);
}
