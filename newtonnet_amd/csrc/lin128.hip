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
//   * the k index is permuted (lane half h owns k in [64h, 64h+64)) identically for A and B so both operands are
//     16-byte vector accesses;
//   * SiLU / SiLU' / bias / accumulate are fused as prologue / epilogue so activations never make an extra
//     HBM round trip.
#include "common.h"

#define W_LD 132  // padded LDS row (floats)
#define LIN_LDS_BYTES (NF * W_LD * 4)

// C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
template <int EPI, bool CHECK>
__device__ __forceinline__ void lin_epilogue(const f32x16 (&acc)[4], const LinGroup& G, const LinArgs& p, int row0,
                                             int r, int h) {
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int cc = nt * 32 + r;
    float bias = 0.f;
    if (EPI == EPI_BIAS) bias = G.bias[cc];
    float* cbase = G.C + (size_t)(row0 + 4 * h) * p.ldc + cc;
    const float* hbase = (EPI == EPI_DSILU) ? G.H + (size_t)(row0 + 4 * h) * p.ldh + cc : nullptr;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int dr = (k & 3) + 8 * (k >> 2);
      if (!CHECK || row0 + 4 * h + dr < p.M) {
        float v = acc[nt][k];
        if (EPI == EPI_BIAS) v += bias;
        if (EPI == EPI_DSILU) v *= dsilu_f(hbase[(size_t)dr * p.ldh]);
        float* cp = cbase + (size_t)dr * p.ldc;
        if (EPI == EPI_ACC) v += *cp;
        *cp = v;
      }
    }
  }
}

template <int PRO, int EPI>
__global__ void __launch_bounds__(256, 2) lin128_kernel(const LinArgs p) {
  extern __shared__ __attribute__((aligned(16))) float wlds[];
  const LinGroup G = p.g[blockIdx.y];

  // stage W (coalesced 16-B loads; each 8-lane group writes one contiguous 128-B run of an LDS row)
  for (int idx = threadIdx.x; idx < NF * (NF / 4); idx += 256) {
    const int n = idx >> 5, k4 = idx & 31;
    const float4 v = reinterpret_cast<const float4*>(G.W)[idx];
    *reinterpret_cast<float4*>(&wlds[n * W_LD + k4 * 4]) = v;
  }
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int n_tiles = (p.M + 31) >> 5;
  const float* wrow = &wlds[r * W_LD + 64 * h];

  for (int tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += gridDim.x * 4) {
    const int row0 = tile << 5;
    const int arow = min(row0 + r, p.M - 1);
    const float4* ap = reinterpret_cast<const float4*>(G.A + (size_t)arow * p.lda + 64 * h);
    float4 a[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) a[t] = ap[t];
    if (PRO == PRO_SILU) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        a[t].x = silu_f(a[t].x);
        a[t].y = silu_f(a[t].y);
        a[t].z = silu_f(a[t].z);
        a[t].w = silu_f(a[t].w);
      }
    }
    f32x16 acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[nt][k] = 0.f;

    // B fragments are double-buffered one k-group ahead; the scheduling barrier keeps hipcc from hoisting all
    // 64 ds_read_b128 to the top (which spills: 256 live VGPRs).
    float4 b[2][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) b[0][nt] = *reinterpret_cast<const float4*>(wrow + nt * 32 * W_LD);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int cur = t & 1, nxt = cur ^ 1;
      if (t < 15) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          b[nxt][nt] = *reinterpret_cast<const float4*>(wrow + nt * 32 * W_LD + 4 * (t + 1));
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].x, b[cur][nt].x, acc[nt], 0, 0, 0);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].y, b[cur][nt].y, acc[nt], 0, 0, 0);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].z, b[cur][nt].z, acc[nt], 0, 0, 0);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].w, b[cur][nt].w, acc[nt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }

    if (row0 + 32 <= p.M)  // wave-uniform: only the last tile pays for per-row predicates
      lin_epilogue<EPI, false>(acc, G, p, row0, r, h);
    else
      lin_epilogue<EPI, true>(acc, G, p, row0, r, h);
  }
}

template <int PRO, int EPI>
static int launch_lin_t(const LinArgs& a, int groups, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    HIP_TRY(hipFuncSetAttribute((const void*)lin128_kernel<PRO, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                LIN_LDS_BYTES));
    attr_set = true;
  }
  const int n_tiles = (a.M + 31) / 32;
  int blocks = cdiv(n_tiles, 4);
  const int cap = 512 / groups;  // 2 resident workgroups per CU x 256 CUs
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
