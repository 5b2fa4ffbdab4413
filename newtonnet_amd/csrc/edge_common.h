// Device helpers shared by the edge kernels (edge.hip: the inference sweeps; train.hip: the tangent sweeps of training).
#pragma once
#include <type_traits>

#include "common.h"

#define ROWS_PER_BLOCK 4  // 4 waves = 256 threads
#ifndef EDGE_ROWS
#define EDGE_ROWS 4       // receiver rows (waves) per workgroup of the four edge kernels (2 and 8 measured: no difference)
#endif

// Cache-policy switches (streaming = non-temporal), kept for A/B timing (-DEDGE_NT_x=1).  Measured on config 2: streaming
// the non-owning endpoint's read of a pair row COSTS 10-20 % in force_fwd / force_bwd / msg_bwd (those reads do hit in L2
// often enough), and streaming the msg store only moves time from msg_fwd to the MLP kernel -- so all four are off.  (The
// one hint that pays is in mlp128.hip: the hidden pre-activations, written once and read once much later.)
#ifndef EDGE_NT_GH
#define EDGE_NT_GH 1    // g_phi rows written by force_bwd (read once by the MLP adjoint)
#endif
#ifndef EDGE_NT_MSG
#define EDGE_NT_MSG 0
#endif
#ifndef EDGE_NT_PHI_FWD
#define EDGE_NT_PHI_FWD 0
#endif
#ifndef EDGE_NT_PHI_BWD
#define EDGE_NT_PHI_BWD 0
#endif
#ifndef EDGE_NT_GMSG
#define EDGE_NT_GMSG 0
#endif
template <bool NT>
__device__ __forceinline__ float4 ld4p(const float* p) {
  return NT ? ld4_nt(p) : ld4(p);
}

// Compile-time ablations for tools/ablate_edge.sh (WRONG results; the shipped build defines none of them):
//   EDGE_ABL_SELF   gather the sender rows from row i instead of j (always cache-hot)
//   EDGE_ABL_PAIR   read the pair rows (phi / g_msg) from row i & 1023 instead of pid[e]
//   EDGE_ABL_TABLE  read the filter table at row 0
//   EDGE_ABL_STORE  drop the pair-row stores
#ifdef EDGE_ABL_SELF
#define ABL_J(j, i) (i)
#else
#define ABL_J(j, i) (j)
#endif
#ifdef EDGE_ABL_PAIR
#define ABL_P(p, i) ((i) & 1023)
#else
#define ABL_P(p, i) (p)
#endif
#ifdef EDGE_ABL_TABLE
#define ABL_G(g) ((g) & 0)
#else
#define ABL_G(g) (g)
#endif
#ifdef EDGE_ABL_STORE
#define ABL_ST(c) ((c) && n_atoms < 0)
#else
#define ABL_ST(c) (c)
#endif

__device__ __forceinline__ int wave_row(int n_rows_padded_blocks) {
#ifdef EDGE_NO_XCD_MAP   // tooling: A/B the XCD-aware block -> row-range map
  const int tile = blockIdx.x;
#else
  const int tile = xcd_tile(blockIdx.x, n_rows_padded_blocks);
#endif
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  return tile * EDGE_ROWS + wave;
}

// Radial filter eps_e = W_e rbf(x_e) (message_edgepart, newtonnet.py:186,210) and d eps_e/dx by cubic interpolation of
// per-layer table planes T[g][f] = eps_f(x_g), D[g][f] = eps_f'(x_g) (graph.hip:filter_table_kernel, FT_G = 2048 intervals,
// built in fp64 on every call).  Evaluating the 20-term contraction per (edge, feature) on the VALU was the
// bottleneck of both message kernels (80 FMA + 40 scalar loads per edge in the adjoint); the tables turn it into
// coalesced row reads from L2 and a handful of FMAs.  Interpolation error ~ h^4 |d4f/dx4| / 24 ~ 2e-8 relative
// (h = 1/2048, fourth derivative ~ (20 pi)^4): below the fp32 rounding of the table entries themselves -- measured
// against fp64 evaluation the value / derivative errors are 4e-8 / 5e-8 of the maximum at 2048 and at 4096 intervals
// alike (1.4e-7 / 2.5e-7 at 1024), so 2048 is the smallest table that costs nothing; it keeps T + D of a layer at 2 MB.
struct FilterW {
  float w[4];   // value weights at nodes -1, 0, 1, 2
};
__device__ __forceinline__ FilterW filter_weights(float u) {
  FilterW f;
  const float um1 = u - 1.f, um2 = u - 2.f, up1 = u + 1.f;
  f.w[0] = -u * um1 * um2 * (1.f / 6.f);
  f.w[1] = up1 * um1 * um2 * 0.5f;
  f.w[2] = -up1 * u * um2 * 0.5f;
  f.w[3] = up1 * u * um1 * (1.f / 6.f);
  return f;
}
__device__ __forceinline__ float4 filter_value(const float* __restrict__ table, int g0, int c4, const FilterW& fw) {
  float4 t[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) t[k] = ld4(table + (size_t)(g0 + k) * NF + c4);
  float4 eps = mul4(t[0], fw.w[0]);
#pragma unroll
  for (int k = 1; k < 4; ++k) eps = fma4(t[k], fw.w[k], eps);
  return eps;
}
// value and derivative: the derivative plane follows the value plane
__device__ __forceinline__ void filter_value_deriv(const float* __restrict__ table, int g0, int c4, const FilterW& fw,
                                                   float4& eps, float4& deps) {
  const float* dt = table + (size_t)FT_ROWS * NF;
  float4 t[4], d[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    t[k] = ld4(table + (size_t)(g0 + k) * NF + c4);
    d[k] = ld4(dt + (size_t)(g0 + k) * NF + c4);
  }
  eps = mul4(t[0], fw.w[0]);
  deps = mul4(d[0], fw.w[0]);
#pragma unroll
  for (int k = 1; k < 4; ++k) {
    eps = fma4(t[k], fw.w[k], eps);
    deps = fma4(d[k], fw.w[k], deps);
  }
}

// First edge of row i whose sender is above i (cols ascend within a row: [beg, mid) are the pairs owned by the other
// endpoint, [mid, end) the pairs this row owns).  One coalesced read of the row's cols per 64 edges.
__device__ __forceinline__ int row_mid(const int* __restrict__ col, int beg, int end, int i, int lane) {
  int mid = beg;
  for (int e = beg; e < end; e += 64) {
    const bool below = (e + lane < end) && col[e + lane] < i;
    mid += __popcll(__ballot(below));
  }
  return mid;
}

