// Dense 128->128 linears on the matrix cores (gfx950, fp32 MFMA).
//
// Replaces the nn.Linear calls on the hot path where the contraction really is a GEMM:
//   message_nodepart        newtonnet/models/newtonnet.py:181-185,209   (M = n_atoms)
//   equiv_message1/2        :188-197,218,222                             (M = n_edges, ~87 % of all FLOPs)
//   equiv_update            :199,230                                     (M = 3 n_atoms)
//   EnergyOutput.layers     newtonnet/models/output.py:90-95             (M = n_atoms)
// and the matching x W products of the reverse sweep (with pre-transposed weights).
//
//   C[M,128] = epilogue( prologue(A)[M,128] . W^T ),  W = [128 out][128 in] row-major (nn.Linear layout)
//
// Design for CDNA4: v_mfma_f32_32x32x2_f32 is exact fp32 (bitwise an fmaf chain) at 64 FLOP/clk/SIMD.
//   * the whole weight matrix (64 KiB, +16 B row pad -> conflict-free ds_read_b128) lives in LDS for the
//     lifetime of a persistent workgroup; 2 workgroups / CU (135 of 160 KiB LDS) so each SIMD holds two waves:
//     one streams its next A tile / writes its C tile while the other issues MFMAs;
//   * one wave owns a 32-row x 128-col output strip: A fragment (32 rows x 128 k) is loaded once into 64 VGPRs,
//     4 independent 32x32 accumulators (64 AGPR/VGPRs) keep the matrix pipe back-to-back;
//   * the k index is permuted (lane half h owns k = 8t + 4h + {0..3}, t = 0..15) identically for A and B, so both
//     operands are 16-byte vector accesses and the two halves of a row read one contiguous 32-byte run per load;
//   * SiLU / SiLU' / bias / accumulate are fused as prologue / epilogue so activations never make an extra
//     HBM round trip.
#include <stdlib.h>
#include <string.h>

#include "common.h"

// compile-time ablation switches for tools/bench_lin.py (all 0 in the shipped library)
#ifndef LIN_ABLATE_NO_LOAD
#define LIN_ABLATE_NO_LOAD 0
#endif
#ifndef LIN_ABLATE_NO_STORE
#define LIN_ABLATE_NO_STORE 0
#endif
#ifndef LIN_ABLATE_NO_MFMA
#define LIN_ABLATE_NO_MFMA 0
#endif
#ifndef LIN_ABLATE_NO_LDS
#define LIN_ABLATE_NO_LDS 0
#endif

#define W_LD 132  // padded LDS row (floats)
#define LIN_LDS_BYTES (NF * W_LD * 4)

// C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
// `pre` holds the epilogue operand (H for EPI_DSILU, the old C for EPI_ACC) in the same layout; it was fetched
// before the MFMA loop so its latency is hidden and no load has to wait behind a (possibly aliasing) store.
template <int EPI, bool CHECK>
__device__ __forceinline__ void lin_epilogue_nt(const f32x16& acc, const float (&pre)[16], const LinGroup& G,
                                                const LinArgs& p, int row0, int r, int h, int nt) {
  const int cc = nt * 32 + r;
  float bias = 0.f;
  if (EPI == EPI_BIAS) bias = G.bias[cc];
  float* cbase = G.C + (size_t)(row0 + 4 * h) * p.ldc + cc;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int dr = (k & 3) + 8 * (k >> 2);
    if (!CHECK || row0 + 4 * h + dr < p.M) {
      float v = acc[k];
      if (EPI == EPI_BIAS) v += bias;
      if (EPI == EPI_DSILU) v *= (p.act == NNHIP_ACT_SILU) ? dsilu_f(pre[k]) : dact_f(pre[k], p.act);
      if (EPI == EPI_ACC) v += pre[k];
      cbase[(size_t)dr * p.ldc] = v;
    }
  }
}

template <int EPI>
__device__ __forceinline__ void lin_prefetch_nt(float (&pre)[16], const LinGroup& G, const LinArgs& p, int row0, int r,
                                                int h, int nt) {
  if (EPI == EPI_DSILU || EPI == EPI_ACC) {
    const float* src = (EPI == EPI_DSILU) ? G.H : G.C;
    const int ld = (EPI == EPI_DSILU) ? p.ldh : p.ldc;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int rr = min(row0 + 4 * h + (k & 3) + 8 * (k >> 2), p.M - 1);  // clamped: tail rows are not stored
      pre[k] = src[(size_t)rr * ld + nt * 32 + r];
    }
  }
}

__device__ __forceinline__ void lin_load_a(float4 (&a)[16], const float* A, int lda, int row, int h) {
  const float4* ap = reinterpret_cast<const float4*>(A + (size_t)row * lda + 4 * h);
#pragma unroll
  for (int t = 0; t < 16; ++t) a[t] = ap[2 * t];
}
__device__ __forceinline__ void lin_silu_a(float4 (&a)[16], const int act) {
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    if (act == NNHIP_ACT_SILU) {
      a[t].x = silu_f(a[t].x);
      a[t].y = silu_f(a[t].y);
      a[t].z = silu_f(a[t].z);
      a[t].w = silu_f(a[t].w);
    } else {
      a[t].x = act_f(a[t].x, act);
      a[t].y = act_f(a[t].y, act);
      a[t].z = act_f(a[t].z, act);
      a[t].w = act_f(a[t].w, act);
    }
  }
}

template <int PRO, int EPI>
__global__ void __launch_bounds__(256, 2) lin128_kernel(const LinArgs p) {
  extern __shared__ __attribute__((aligned(16))) float wlds[];
  const LinGroup G = p.g[blockIdx.y];

  // stage W (coalesced 16-B loads; each 8-lane group writes one contiguous 128-B run of an LDS row)
  {  // all 16 requests in flight before the first LDS write (a rolled load -> store loop serializes 16 L2 round trips)
    float4 wv[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) wv[q] = reinterpret_cast<const float4*>(G.W)[threadIdx.x + 256 * q];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int idx = threadIdx.x + 256 * q;
      const int n = idx >> 5, k4 = idx & 31;
      *reinterpret_cast<float4*>(&wlds[n * W_LD + k4 * 4]) = wv[q];
    }
  }
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int n_tiles = (p.M + 31) >> 5;
  const float* wrow = &wlds[r * W_LD + 4 * h];

  // Software pipeline across tiles: the NEXT tile's A fragment is requested at the top of the current tile, so it
  // lands under this tile's 256 MFMAs, and it is queued BEFORE this tile's stores (vmcnt retires in order: a load
  // queued behind stores would wait for their acknowledgements too).  Column blocks are processed one after the
  // other (one live 32x32 accumulator; a dependent 32x32x2 chain issues back-to-back at 64 cycles), so a block's
  // epilogue operand / stores overlap the next block's MFMAs and the register budget leaves room for the
  // double-buffered A fragment.
  int tile = blockIdx.x * 4 + wave;
  const int tile_step = gridDim.x * 4;
  if (tile >= n_tiles) return;
  float4 a[16], a_next[16];
  lin_load_a(a, G.A, p.lda, min((tile << 5) + r, p.M - 1), h);
  // Drain the prologue loads here, so that inside the loop the A fragment is always a plain register value:
  // hipcc's waitcnt pass then puts no vmcnt wait in front of the MFMAs (an in-order vmcnt wait there would also
  // wait for the previous tile's 64 stores), only one at the a <- a_next hand-off at the end of a tile.
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  for (; tile < n_tiles; tile += tile_step) {
    const int row0 = tile << 5;
    // unconditional (clamped) prefetch: a branch here would make the pass assume the shorter queue
    if (!LIN_ABLATE_NO_LOAD) lin_load_a(a_next, G.A, p.lda, min((min(tile + tile_step, n_tiles - 1) << 5) + r, p.M - 1), h);
    if (PRO == PRO_SILU) lin_silu_a(a, p.act);
    const bool full = row0 + 32 <= p.M;  // wave-uniform: only the last tile pays for per-row predicates
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      float pre[16];
      lin_prefetch_nt<EPI>(pre, G, p, row0, r, h, nt);
      f32x16 acc;
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[k] = 0.f;
      const float* wnt = wrow + nt * 32 * W_LD;
      float4 b = *reinterpret_cast<const float4*>(wnt);  // k-slots 8t + 4h + {0..3}
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        float4 bn;
        if (t < 15 && !LIN_ABLATE_NO_LDS) bn = *reinterpret_cast<const float4*>(wnt + 8 * (t + 1));
        if (LIN_ABLATE_NO_LDS) bn = b;
        if (!LIN_ABLATE_NO_MFMA) {
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].x, b.x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].y, b.y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].z, b.z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].w, b.w, acc, 0, 0, 0);
        } else {
          acc[0] += a[t].x * b.x + a[t].y * b.y + a[t].z * b.z + a[t].w * b.w;
        }
        if (t < 15) b = bn;
      }
      __builtin_amdgcn_sched_barrier(0);
      if (nt == 3) {
        // hand-off before the last block's stores are queued: the wait for a_next then only has to see the first
        // three blocks' stores retire (issued >= 4096 cycles ago), and nothing waits at the top of the next tile
        // (asm volatile pins the copy here: as plain assignments hipcc sinks it below the stores, behind vmcnt(0))
#pragma unroll
        for (int t = 0; t < 16; ++t)
          asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                       : "=&v"(a[t].x), "=&v"(a[t].y), "=&v"(a[t].z), "=&v"(a[t].w)
                       : "v"(a_next[t].x), "v"(a_next[t].y), "v"(a_next[t].z), "v"(a_next[t].w));
      }
      if (LIN_ABLATE_NO_STORE && acc[0] != 12345.678f) continue;
      if (full)
        lin_epilogue_nt<EPI, false>(acc, pre, G, p, row0, r, h, nt);
      else
        lin_epilogue_nt<EPI, true>(acc, pre, G, p, row0, r, h, nt);
    }
  }
}

static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

template <int PRO, int EPI>
static int launch_lin_t(const LinArgs& a, int groups, hipStream_t s) {
  static const int block_cap = env_int("NNHIP_LIN_BLOCKS", 512);  // tuning knob (tools/bench_lin.py)
  // one-time kernel attribute; a function-local static is initialised exactly once even with concurrent host threads
  static const hipError_t attr_rc = hipFuncSetAttribute((const void*)lin128_kernel<PRO, EPI>,
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, LIN_LDS_BYTES);
  HIP_TRY(attr_rc);
  const int n_tiles = (a.M + 31) / 32;
  int blocks = cdiv(n_tiles, 4);
  const int cap = block_cap / groups;  // default 512: 2 resident workgroups per CU x 256 CUs
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  lin128_kernel<PRO, EPI><<<dim3(blocks, groups), 256, LIN_LDS_BYTES, s>>>(a);
  LAUNCH_CHECK();
  return 0;
}

// groups = 1 or 2 (two independent linears of the same shape in one launch: blockIdx.y selects)
int launch_lin(int pro, int epi, const LinArgs& a, int groups, hipStream_t s) {
  if (a.M <= 0) return 0;
  ScopedTimer t0(TC_LIN, s);
  ScopedTimer t1(TC_LIN1, s);
#define CASE(P, E) \
  if (pro == P && epi == E) return launch_lin_t<P, E>(a, groups, s);
  CASE(PRO_NONE, EPI_STORE)
  CASE(PRO_NONE, EPI_BIAS)
  CASE(PRO_NONE, EPI_DSILU)
  CASE(PRO_NONE, EPI_ACC)
  CASE(PRO_SILU, EPI_STORE)
  CASE(PRO_SILU, EPI_BIAS)
#undef CASE
  nnhip_set_error("launch_lin: unsupported prologue/epilogue %d/%d", pro, epi);
  return NNHIP_E_INVALID;
}

int launch_lin_wide(const float* X, int ldx, const float* W, float* Y, int ldy, int M, bool acc, hipStream_t s);   // node128.hip

// C ABI: one dense 128->128 linear (see include/newtonnet_hip.h)
extern "C" int nnhip_linear128(const float* A, int32_t lda, const float* W, float* C, int32_t ldc, const float* bias,
                               const float* H, int32_t ldh, int32_t M, int32_t prologue, int32_t epilogue,
                               void* stream) {
  if (!A || !W || !C || M < 0 || lda < NF || ldc < NF || (epilogue == EPI_BIAS && !bias) ||
      (epilogue == EPI_DSILU && (!H || ldh < NF))) {
    nnhip_set_error("nnhip_linear128: bad arguments");
    return NNHIP_E_INVALID;
  }
  // small products without prologue / bias: the row-local form (node128.hip), no weight staging
  if (prologue == PRO_NONE && (epilogue == EPI_STORE || epilogue == EPI_ACC) && M <= 32 * 1536)
    return launch_lin_wide(A, lda, W, C, ldc, M, epilogue == EPI_ACC, (hipStream_t)stream);
  LinArgs a;
  memset(&a, 0, sizeof(a));
  a.g[0] = {A, W, C, bias, H};
  a.M = M;
  a.lda = lda;
  a.ldc = ldc;
  a.ldh = ldh;
  return launch_lin(prologue, epilogue, a, 1, (hipStream_t)stream);
}
