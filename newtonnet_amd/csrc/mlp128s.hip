// Fused two-layer 128->128->128 edge MLP and its adjoint with SPLIT-f16 products on the matrix cores (gfx950).
//
// Same contract, same tiling and the same register hand-over between the two GEMMs as mlp128.hip (read that header first); what
// changes is how a product a*b of two fp32 numbers reaches the accumulator.  v_mfma_f32_32x32x2_f32 runs at 1/16 of the
// f16 rate, and the edge MLPs are the MFMA-bound 43 % of the inference step.  Here every fp32 operand is written as
//     v * S = hi + lo,   hi = f16(v * S),   lo = f16(v * S - hi)          (S a power of two: exact)
// i.e. two f16 numbers that carry 22 significant bits of v, and  a*b  is accumulated in fp32 as
//     hi_a hi_b + hi_a lo_b + lo_a hi_b                                   (3 x v_mfma_f32_32x32x16_f16)
// The dropped terms (lo_a lo_b and the two 2^-22 representation residues) are below 3 * 2^-22 |a b|; an fp32 product chain
// rounds 64 times per output at 2^-24, this one 24 times, so the end-to-end error against the fp64 oracle is the same to
// within noise (tools/bench_mlp.py prints both; DESIGN.md section 5 has the numbers).  Three 8-pass MFMAs replace eight 16-pass
// ones per 16 k-values: 5.3x less matrix-pipe time, and the kernel becomes bound by its HBM traffic.
//
// Scales.  f16 has 5 exponent bits, so operands are scaled into [2^14, 2^15) by their row's largest magnitude before the
// split: activations per ROW (a lane owns one pair row: its own 64 values + one cross-half exchange), weights per MATRIX (a
// workgroup-wide maximum while the matrix is staged into LDS).  Both scales are powers of two: applying and removing them is
// exact, and rows / matrices of any magnitude keep their 22 bits.  (The bound is relative to the row's and the matrix's LARGEST
// element -- |error| <~ 2^-21 max|row| max|W| sqrt(K) per output, the class of the fp32 rounding of the dominant products;
// elements far below their row's maximum are not resolved to 22 bits of their own.)
//
// LDS image of a weight matrix: two f16 planes (hi, lo) of [128 outputs][128 k-slots], 272-byte row pitch (conflict-free
// ds_read_b128), with the k-slots permuted so that the 8 slots a lane feeds to one MFMA are contiguous AND match what that
// lane already holds of the activations (features 16T + 4h + {0..3} and 16T + 8 + 4h + {0..3} for MFMA T, lane half h): the
// global loads of X and the stage-1 -> stage-2 register hand-over are exactly those of mlp128.hip.
#include <stdlib.h>
#include <string.h>

#include "common.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef __bf16 b4 __attribute__((ext_vector_type(4)));

#define SW_PITCH 272                         // bytes per output row of one plane (128 f16 + 16 pad)
#define SW_PLANE (NF * SW_PITCH)             // 34 816
#define SW_MAT (2 * SW_PLANE)                // hi plane, lo plane
#define MLPS_WAVES 8
#define MLPS_THREADS (64 * MLPS_WAVES)
#define MLPS_LDS_BYTES (2 * SW_MAT + 2 * MLPS_WAVES * 4)   // two matrices + the scratch of the matrix-maximum reduction

#ifndef MLP_NT_H
#define MLP_NT_H 1
#endif
#ifndef MLP_NT_HL
#define MLP_NT_HL 1
#endif

// S = 2^(14 - floor(log2 m)) as (S, 1/S); (1, 1) for zero / tiny / non-finite m
__device__ __forceinline__ void pow2_scale(float m, float& S, float& inv) {
  const int e = (int)((__float_as_uint(m) >> 23) & 0xffu);
  const bool ok = e >= 40 && e < 255;
  S = ok ? __uint_as_float((unsigned)(268 - e) << 23) : 1.0f;
  inv = ok ? __uint_as_float((unsigned)(e - 14) << 23) : 1.0f;
}
__device__ __forceinline__ float amax4(float m, const float4& v) {
  return fmaxf(fmaxf(fmaxf(m, fabsf(v.x)), fmaxf(fabsf(v.y), fabsf(v.z))), fabsf(v.w));
}
// the two f16 pieces of 8 scaled values
__device__ __forceinline__ void split8(const float (&v)[8], float S, h8& hi, h8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float s = v[j] * S;
    const _Float16 a = (_Float16)s;
    hi[j] = a;
    lo[j] = (_Float16)(s - (float)a);
  }
}
__device__ __forceinline__ void split4(const float4& v, float S, h4& hi, h4& lo) {
  const float s[4] = {v.x * S, v.y * S, v.z * S, v.w * S};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const _Float16 a = (_Float16)s[j];
    hi[j] = a;
    lo[j] = (_Float16)(s[j] - (float)a);
  }
}

__device__ __forceinline__ void mlps_load_x(float4 (&x)[16], const float* X, int ldx, int row, int h) {
#ifdef MLPS_ABL_X   // tooling (wrong results): every tile reads the first 32 rows -- the price of streaming X from HBM
  row &= 31;
#endif
  const float4* xp = reinterpret_cast<const float4*>(X + (size_t)row * ldx + 4 * h);
#pragma unroll
  for (int t = 0; t < 16; ++t) x[t] = xp[2 * t];   // features 8t + 4h + {0..3}
}

// Stage W1 | W2 of one MLP: request (16 float4 per thread), matrix maxima, split, write the permuted planes.
// (A macro: the fragments must stay in registers between the request and the commit.)
#define MLPS_W_LD(q, W1p, W2p)                                                          \
  const float4 w1v##q = reinterpret_cast<const float4*>(W1p)[threadIdx.x + 512 * q];     \
  const float4 w2v##q = reinterpret_cast<const float4*>(W2p)[threadIdx.x + 512 * q];
#define MLPS_W_REQUEST(W1p, W2p)                                                                          \
  MLPS_W_LD(0, W1p, W2p) MLPS_W_LD(1, W1p, W2p) MLPS_W_LD(2, W1p, W2p) MLPS_W_LD(3, W1p, W2p)               \
  MLPS_W_LD(4, W1p, W2p) MLPS_W_LD(5, W1p, W2p) MLPS_W_LD(6, W1p, W2p) MLPS_W_LD(7, W1p, W2p)               \
  __builtin_amdgcn_sched_barrier(0);
#define MLPS_W_MAX(q) \
  m1 = amax4(m1, w1v##q); \
  m2 = amax4(m2, w2v##q);
#define MLPS_W_ST(q)                                                                                      \
  {                                                                                                       \
    const int idx = threadIdx.x + 512 * q, o = idx >> 5, c = idx & 31;                                    \
    const int off = o * SW_PITCH + 2 * ((c >> 2) * 16 + (c & 1) * 8 + ((c >> 1) & 1) * 4);                \
    if (bf) { /* bf16 compute mode: one unscaled bf16 plane per matrix */                                 \
      *reinterpret_cast<b4*>(img + off) = to_b4(w1v##q);                                                  \
      *reinterpret_cast<b4*>(img + SW_MAT + off) = to_b4(w2v##q);                                         \
    } else {                                                                                              \
      h4 hi, lo;                                                                                          \
      split4(w1v##q, sw1, hi, lo);                                                                        \
      *reinterpret_cast<h4*>(img + off) = hi;                                                             \
      *reinterpret_cast<h4*>(img + SW_PLANE + off) = lo;                                                  \
      split4(w2v##q, sw2, hi, lo);                                                                        \
      *reinterpret_cast<h4*>(img + SW_MAT + off) = hi;                                                    \
      *reinterpret_cast<h4*>(img + SW_MAT + SW_PLANE + off) = lo;                                         \
    }                                                                                                     \
  }
// (ends with the images visible to every wave; iw1 / iw2 = inverse scales of the two matrices)
#define MLPS_W_COMMIT()                                                                                   \
  {                                                                                                       \
    float m1 = 0.f, m2 = 0.f;                                                                             \
    MLPS_W_MAX(0) MLPS_W_MAX(1) MLPS_W_MAX(2) MLPS_W_MAX(3) MLPS_W_MAX(4) MLPS_W_MAX(5) MLPS_W_MAX(6) MLPS_W_MAX(7) \
    _Pragma("unroll") for (int d = 32; d >= 1; d >>= 1) {                                                 \
      m1 = fmaxf(m1, __shfl_xor(m1, d));                                                                  \
      m2 = fmaxf(m2, __shfl_xor(m2, d));                                                                  \
    }                                                                                                     \
    if (lane == 0) {                                                                                      \
      red[2 * wave] = m1;                                                                                 \
      red[2 * wave + 1] = m2;                                                                             \
    }                                                                                                     \
    __syncthreads();                                                                                      \
    _Pragma("unroll") for (int w_ = 0; w_ < MLPS_WAVES; ++w_) {                                           \
      m1 = fmaxf(m1, red[2 * w_]);                                                                        \
      m2 = fmaxf(m2, red[2 * w_ + 1]);                                                                    \
    }                                                                                                     \
    float sw1, sw2;                                                                                       \
    pow2_scale(m1, sw1, iw1);                                                                             \
    pow2_scale(m2, sw2, iw2);                                                                             \
    if (bf) iw1 = iw2 = 1.0f;                                                                             \
    MLPS_W_ST(0) MLPS_W_ST(1) MLPS_W_ST(2) MLPS_W_ST(3) MLPS_W_ST(4) MLPS_W_ST(5) MLPS_W_ST(6) MLPS_W_ST(7) \
    __syncthreads();                                                                                      \
  }

__device__ __forceinline__ b4 to_b4(const float4& v) {
  b4 w;
  w[0] = (__bf16)v.x, w[1] = (__bf16)v.y, w[2] = (__bf16)v.z, w[3] = (__bf16)v.w;
  return w;
}
// bf16 compute mode: the same block from ONE bf16 plane per operand (bh holds the bf16 bits), one MFMA per 16 k-values
__device__ __forceinline__ f32x16 bf16_block(const char* wrow, const h8 (&bh)[8]) {
  f32x16 acc;
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.f;
  b8 a0 = *reinterpret_cast<const b8*>(wrow), a1 = *reinterpret_cast<const b8*>(wrow + 32);
#pragma unroll
  for (int T = 0; T < 8; ++T) {
    b8 a2;
    if (T < 6) a2 = *reinterpret_cast<const b8*>(wrow + 32 * (T + 2));
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, __builtin_bit_cast(b8, bh[T]), acc, 0, 0, 0);
    a0 = a1;
    if (T < 6) a1 = a2;
  }
  return acc;
}
__device__ __forceinline__ h8 bf16_bits8(const float (&v)[8]) {
  b8 w;
#pragma unroll
  for (int j = 0; j < 8; ++j) w[j] = (__bf16)v[j];
  return __builtin_bit_cast(h8, w);
}
// one 32-feature block of D^T = W . B^T from the split operands: 8 MFMA triples, A fragments two triples ahead
__device__ __forceinline__ f32x16 split_block(const char* wrow, const h8 (&bh)[8], const h8 (&bl)[8]) {
  f32x16 acc;
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.f;
  h8 ah0 = *reinterpret_cast<const h8*>(wrow), al0 = *reinterpret_cast<const h8*>(wrow + SW_PLANE);
  h8 ah1 = *reinterpret_cast<const h8*>(wrow + 32), al1 = *reinterpret_cast<const h8*>(wrow + SW_PLANE + 32);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int T = 0; T < 8; ++T) {
    h8 ah2, al2;
    if (T < 6) {
      ah2 = *reinterpret_cast<const h8*>(wrow + 32 * (T + 2));
      al2 = *reinterpret_cast<const h8*>(wrow + SW_PLANE + 32 * (T + 2));
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bh[T], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bl[T], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0, bh[T], acc, 0, 0, 0);
    ah0 = ah1;
    al0 = al1;
    if (T < 6) {
      ah1 = ah2;
      al1 = al2;
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
  return acc;
}

// The next tile's X request (a macro: placed at one of two points of the tile loop)
#define MLPS_PREFETCH_X()                                                                  \
  {                                                                                        \
    const bool last = tile + tile_step >= n_tiles;                                         \
    const float* xn = (last && more) ? P.a[1].X : p.X;                                     \
    const int ldn = (last && more) ? P.a[1].ldx : p.ldx;                                   \
    const int nt = last ? tile0 : tile + tile_step;                                        \
    mlps_load_x(x, xn, ldn, min((min(nt, n_tiles - 1) << 5) + r, M - 1), h);               \
    __builtin_amdgcn_sched_barrier(0);                                                     \
  }
#ifndef MLPS_EARLY_X
#define MLPS_EARLY_X 0   // 1: forward mode requests the next X before stage 1 (a whole tile of cover; tooling A/B)
#endif

template <int MODE, bool ACCUM_LAST>
__global__ void __launch_bounds__(MLPS_THREADS, MLPS_WAVES / 4) mlp128s_kernel(const MlpPair P) {
  extern __shared__ __attribute__((aligned(16))) char img[];
  float* red = reinterpret_cast<float*>(img + 2 * SW_MAT);

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int M = mlp_rows(P.a[0]);
  if (M <= 0) return;   // (uniform; only possible with a device-side count)
  const bool bf = P.a[0].bf16 != 0;   // (uniform) bf16 compute mode: one bf16 plane per operand, no scales, one MFMA per k-group
  const int n_tiles = (M + 31) >> 5;
  const char* w1row = img + r * SW_PITCH + 16 * h;             // + nb * 32 * SW_PITCH + 32 T
  const char* w2row = img + SW_MAT + r * SW_PITCH + 16 * h;

  // tile -> wave map of mlp128.hip: SIMDs first, the two waves of a SIMD second
  const int n_simd = gridDim.x * 4;
  const int tile0 = (wave >> 2) * n_simd + blockIdx.x * 4 + (wave & 3);
  const int tile_step = gridDim.x * MLPS_WAVES;
  float4 x[16];
  float iw1, iw2;
  {
    MLPS_W_REQUEST(P.a[0].W1, P.a[0].W2)
    mlps_load_x(x, P.a[0].X, P.a[0].ldx, min((min(tile0, n_tiles - 1) << 5) + r, M - 1), h);
    __builtin_amdgcn_sched_barrier(0);
    MLPS_W_COMMIT()
  }
  for (int ph = 0; ph < P.n; ++ph) {
    struct { const float* X; float* H; float* Y; int ldx, ldh, ldy; } p;
    const bool h_frag = P.a[0].h_frag != 0;
    p.X = ph ? P.a[1].X : P.a[0].X;
    p.H = ph ? P.a[1].H : P.a[0].H;
    p.Y = ph ? P.a[1].Y : P.a[0].Y;
    p.ldx = ph ? P.a[1].ldx : P.a[0].ldx;
    p.ldh = ph ? P.a[1].ldh : P.a[0].ldh;
    p.ldy = ph ? P.a[1].ldy : P.a[0].ldy;
    const bool accum = ACCUM_LAST && ph == P.n - 1;
    const bool more = ph + 1 < P.n;
    if (ph > 0) {
      __syncthreads();                     // every wave is done with the previous phase's images
      MLPS_W_REQUEST(P.a[1].W1, P.a[1].W2)
      MLPS_W_COMMIT()
    }
    for (int tile = tile0; tile < n_tiles; tile += tile_step) {
      const int e = (tile << 5) + r;
      const int ec = min(e, M - 1);
      const bool live = e < M;

      // ---------------- operands of stage 1: this lane's row, scaled by the row maximum, split
      h8 bh[8], bl[8];
      float inv1;
      {
        float m = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) m = amax4(m, x[t]);
        m = fmaxf(m, __shfl_xor(m, 32));
        float S, inv;
        pow2_scale(m, S, inv);
        inv1 = bf ? 1.0f : inv * iw1;
#pragma unroll
        for (int T = 0; T < 8; ++T) {
          const float v[8] = {x[2 * T].x, x[2 * T].y, x[2 * T].z, x[2 * T].w,
                              x[2 * T + 1].x, x[2 * T + 1].y, x[2 * T + 1].z, x[2 * T + 1].w};
          if (bf)
            bh[T] = bf16_bits8(v);
          else
            split8(v, S, bh[T], bl[T]);
        }
      }
      if (MLPS_EARLY_X && MODE == MODE_FWD) MLPS_PREFETCH_X()
      // ---------------- stage 1: H^T = W1 . X^T  (4 blocks of 32 features)
      float hs[4][16];
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        float4 hin[4];
        if (MODE != MODE_FWD) {
          const float4* hp = h_frag ? reinterpret_cast<const float4*>(p.H) + ((size_t)tile * 4 + nb) * 256 + lane
                                    : reinterpret_cast<const float4*>(p.H + (size_t)ec * p.ldh + nb * 32 + 4 * h);
          const int hs4 = h_frag ? 64 : 2;
#pragma unroll
          for (int q = 0; q < 4; ++q) hin[q] = MLP_NT_HL ? ld4_nt(reinterpret_cast<const float*>(hp + hs4 * q)) : hp[hs4 * q];
        }
        f32x16 acc = bf ? bf16_block(w1row + nb * 32 * SW_PITCH, bh) : split_block(w1row + nb * 32 * SW_PITCH, bh, bl);
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] *= inv1;
        if (MODE == MODE_FWD) {
#ifdef MLPS_ABL_NO_H   // tooling (wrong results downstream): the forward without its hidden-tile stores -- what recomputing h in the adjoint could save
          if (false) {
#else
          if (live || h_frag) {
#endif
            float4* hp = h_frag ? reinterpret_cast<float4*>(p.H) + ((size_t)tile * 4 + nb) * 256 + lane
                                : reinterpret_cast<float4*>(p.H + (size_t)e * p.ldh + nb * 32 + 4 * h);
            const int hs4 = h_frag ? 64 : 2;
            // fragment order = private scratch of this forward and its adjoint (inference): keep silu'(h), which shares the
            // sigmoid with the activation here and is all the adjoint wants of h (an exp, a reciprocal and four products per
            // element less in a kernel that is bound by its VALU issue, mlp128r.hip); row-major H is the pre-activation itself
            float keep[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) keep[k] = h_frag ? dsilu_f(acc[k]) : acc[k];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float4 hv = make_float4(keep[4 * q], keep[4 * q + 1], keep[4 * q + 2], keep[4 * q + 3]);
              if (MLP_NT_H)
                st4_nt(reinterpret_cast<float*>(hp + hs4 * q), hv);
              else
                hp[hs4 * q] = hv;
            }
          }
#pragma unroll
          for (int k = 0; k < 16; ++k) hs[nb][k] = silu_f(acc[k]);
        } else if (MODE == MODE_TAN2) {
          // tangent of the adjoint: G = dT act'(H) + T2 act''(H) Hd, kept (row-major) for the weight-gradient products
          const float* t2p = (ph ? P.a[1].T2 : P.a[0].T2) + (size_t)ec * p.ldh + nb * 32 + 4 * h;
          const float* hdp = (ph ? P.a[1].Hd : P.a[0].Hd) + (size_t)ec * p.ldh + nb * 32 + 4 * h;
          float* gp = (ph ? P.a[1].G : P.a[0].G) + (size_t)e * p.ldh + nb * 32 + 4 * h;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 t2 = ld4(t2p + 8 * q), hd = ld4(hdp + 8 * q);
            float4 gv;
            gv.x = fmaf(acc[4 * q], dsilu_f(hin[q].x), t2.x * d2silu_f(hin[q].x) * hd.x);
            gv.y = fmaf(acc[4 * q + 1], dsilu_f(hin[q].y), t2.y * d2silu_f(hin[q].y) * hd.y);
            gv.z = fmaf(acc[4 * q + 2], dsilu_f(hin[q].z), t2.z * d2silu_f(hin[q].z) * hd.z);
            gv.w = fmaf(acc[4 * q + 3], dsilu_f(hin[q].w), t2.w * d2silu_f(hin[q].w) * hd.w);
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
          const bool kept = MODE == MODE_BWD && h_frag;   // (the forward left silu'(h) in the fragment-order scratch)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            hs[nb][4 * q] = acc[4 * q] * (kept ? hin[q].x : dsilu_f(hin[q].x));
            hs[nb][4 * q + 1] = acc[4 * q + 1] * (kept ? hin[q].y : dsilu_f(hin[q].y));
            hs[nb][4 * q + 2] = acc[4 * q + 2] * (kept ? hin[q].z : dsilu_f(hin[q].z));
            hs[nb][4 * q + 3] = acc[4 * q + 3] * (kept ? hin[q].w : dsilu_f(hin[q].w));
          }
        }
      }

      // ---------------- operands of stage 2: the stage-1 register tile (MFMA T takes hs[T >> 1][8 (T & 1) .. + 7])
      float inv2;
      {
        float m = 0.f;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
          for (int k = 0; k < 16; ++k) m = fmaxf(m, fabsf(hs[nb][k]));
        m = fmaxf(m, __shfl_xor(m, 32));
        float S, inv;
        pow2_scale(m, S, inv);
        inv2 = bf ? 1.0f : inv * iw2;
#pragma unroll
        for (int T = 0; T < 8; ++T) {
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = hs[T >> 1][8 * (T & 1) + j];
          if (bf)
            bh[T] = bf16_bits8(v);
          else
            split8(v, S, bh[T], bl[T]);
        }
      }

      // X of the NEXT tile (of this phase, or the first tile of the next phase): requested once the stage-1 tile has been split
      // (the register budget of two waves per SIMD does not hold x, both operand sets and the stage-1 tile at once)
      if (!(MLPS_EARLY_X && MODE == MODE_FWD)) MLPS_PREFETCH_X()

      // ---------------- stage 2: Y^T = W2 . act^T
#pragma unroll
      for (int nb2 = 0; nb2 < 4; ++nb2) {
        const f32x16 acc = bf ? bf16_block(w2row + nb2 * 32 * SW_PITCH, bh) : split_block(w2row + nb2 * 32 * SW_PITCH, bh, bl);
        float4 yold[4];
        if (ACCUM_LAST) {   // (requested behind the MFMA chain: 16 more live registers across it would spill)
          const float4* yp = reinterpret_cast<const float4*>(p.Y + (size_t)ec * p.ldy + nb2 * 32 + 4 * h);
#pragma unroll
          for (int q = 0; q < 4; ++q) yold[q] = accum ? yp[2 * q] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (live) {
          float4* yp = reinterpret_cast<float4*>(p.Y + (size_t)e * p.ldy + nb2 * 32 + 4 * h);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float4 v = make_float4(acc[4 * q] * inv2, acc[4 * q + 1] * inv2, acc[4 * q + 2] * inv2, acc[4 * q + 3] * inv2);
            if (ACCUM_LAST) {
              v.x += yold[q].x;
              v.y += yold[q].y;
              v.z += yold[q].z;
              v.w += yold[q].w;
            }
#ifdef MLPS_NT_Y   // tooling A/B: streaming stores of the stage-2 output (measured 0.90 vs 0.60 ms per step: the 32-byte row pieces are not combined)
            st4_nt(reinterpret_cast<float*>(yp + 2 * q), v);
#else
            yp[2 * q] = v;
#endif
          }
        }
      }
    }
  }
}

template <int MODE, bool ACCUM_LAST>
static int launch_split_t(const MlpPair& a, hipStream_t s) {
  static const hipError_t attr_rc = hipFuncSetAttribute((const void*)mlp128s_kernel<MODE, ACCUM_LAST>,
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, MLPS_LDS_BYTES);
  HIP_TRY(attr_rc);
  const int n_tiles = (a.a[0].M + 31) / 32;
  int blocks = cdiv(n_tiles, MLPS_WAVES);
  if (blocks > 256) blocks = 256;  // one persistent workgroup per CU (136 KiB of LDS each)
  mlp128s_kernel<MODE, ACCUM_LAST><<<blocks, MLPS_THREADS, MLPS_LDS_BYTES, s>>>(a);
  LAUNCH_CHECK();
  return 0;
}

// (mlp128.hip: launch_mlp_dispatch) every mode of the persistent kernel, SiLU
int launch_mlp_split(int mode, bool accum_last, const MlpPair& P, hipStream_t s) {
  if (mode == MODE_FWD && !accum_last) return launch_split_t<MODE_FWD, false>(P, s);
  if (mode == MODE_BWD && !accum_last) return launch_split_t<MODE_BWD, false>(P, s);
  if (mode == MODE_BWD && accum_last) return launch_split_t<MODE_BWD, true>(P, s);
  if (mode == MODE_TAN && !accum_last) return launch_split_t<MODE_TAN, false>(P, s);
  if (mode == MODE_TAN && accum_last) return launch_split_t<MODE_TAN, true>(P, s);
  if (mode == MODE_TAN2 && !accum_last) return launch_split_t<MODE_TAN2, false>(P, s);
  if (mode == MODE_TAN2 && accum_last) return launch_split_t<MODE_TAN2, true>(P, s);
  nnhip_set_error("launch_mlp_split: unsupported mode %d/%d", mode, (int)accum_last);
  return NNHIP_E_INVALID;
}
