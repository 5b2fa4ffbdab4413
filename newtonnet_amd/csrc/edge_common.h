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
//   EDGE_ABL_HALF_TABLE  (edge.hip) no filter-table reads for the pairs the other endpoint owns
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

// Split rows (the four inference kernels of edge.hip): a receiver row's edges are shared by WPR consecutive waves of the
// workgroup (wave w of a row takes the edge pairs 2 w, 2 w + 2 WPR, ... of each of the row's two ranges), so a row lives 1 / WPR as
// long and WPR times fewer rows are in flight at the same occupancy: the second read of a pair row by its other endpoint then
// finds the line still in the XCD's 4 MB L2.  The partial row sums meet in LDS and are added in wave order by
// the row's first wave: a fixed order, deterministic, independent of the order of the molecules.
// WPR (waves per row: 1, 2 or 4) is a template parameter of each kernel; the shipped settings are below (measured on config 2,
// profiles/r04_split_rows_*: two waves per row take 20-30 % of the fabric reads out of force_fwd / msg_bwd and 3-5 % of the time out
// of the message kernels; force_bwd reads six rows of its own per WAVE and is faster unsplit).
#ifndef EDGE_WPR_MSG_FWD
#define EDGE_WPR_MSG_FWD 2
#endif
#ifndef EDGE_WPR_FORCE_FWD
#define EDGE_WPR_FORCE_FWD 2
#endif
#ifndef EDGE_WPR_FORCE_BWD
#define EDGE_WPR_FORCE_BWD 1
#endif
#ifndef EDGE_WPR_MSG_BWD
#define EDGE_WPR_MSG_BWD 2
#endif
template <int WPR>
__device__ __forceinline__ int wave_row_split(int n_blocks, int& part) {
#ifdef EDGE_NO_XCD_MAP
  const int tile = blockIdx.x;
#else
  const int tile = xcd_tile(blockIdx.x, n_blocks);
#endif
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  part = wave % WPR;
  return tile * (EDGE_ROWS / WPR) + wave / WPR;
}
// K float4 row sums per lane, valid in lanes 0-31 (halves already folded): parts 1 .. WPR-1 publish, part 0 adds them in order.
// EVERY wave of the workgroup must call this (a workgroup barrier inside); `comb` = (EDGE_ROWS / WPR) * (WPR - 1) * K * 32 float4.
template <int WPR, int K>
__device__ __forceinline__ void row_combine(float4 (&acc)[K], float4* comb, int part, int lane) {
  if (WPR == 1) return;
  const int row_local = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) / WPR;
  float4* mine = comb + (size_t)row_local * (WPR - 1) * K * 32;
  if (part > 0 && lane < 32) {
#pragma unroll
    for (int k = 0; k < K; ++k) mine[((part - 1) * K + k) * 32 + lane] = acc[k];
  }
  __syncthreads();
  if (part == 0 && lane < 32) {
#pragma unroll
    for (int q = 1; q < WPR; ++q)
#pragma unroll
      for (int k = 0; k < K; ++k) acc[k] = add4(acc[k], mine[((q - 1) * K + k) * 32 + lane]);
  }
}
#define EDGE_COMB_SIZE(WPR, K) ((WPR) > 1 ? (EDGE_ROWS / (WPR)) * ((WPR) - 1) * (K) * 32 : 1)

// Radial filter eps_e = W_e rbf(x_e) (message_edgepart, newtonnet.py:186,210) and d eps_e/dx from per-layer tables
// (graph.hip:filter_table_kernel, FT_G intervals, built in fp64 on every call).  Evaluating the 20-term contraction per
// (edge, feature) on the VALU was the bottleneck of both message kernels (80 FMA + 40 scalar loads per edge in the adjoint); the
// tables turn it into coalesced row reads from L2 and a handful of FMAs.
//
// Three planes over the nodes x_g = g / FT_G (row = g + 1): T[g] = eps(x_g), S[g] = (eps(x_g+1) - eps(x_g)) FT_G (the secant
// slope, formed in fp64 BEFORE the one rounding -- differencing the fp32 values would amplify their rounding by FT_G) and
// D[g] = eps'(x_g).  On the interval [x_g, x_g+1], u in [0, 1), h = 1 / FT_G:
//   * value only (message forward; the adjoint's pairs owned by the other endpoint): 4-point cubic Lagrange on the T plane
//     alone, rows g-1 .. g+2 -- 2 KiB contiguous, and the kernels that need no derivative touch a 1.5 MB plane that stays
//     resident in the 4 MB L2 of an XCD (interleaving the planes made msg_fwd fetch 3.7x more from the fabric);
//   * value AND derivative (the adjoint's own pairs, the tangent kernels of training): cubic Hermite through (T, D) at both ends
//       eps(u)  = T_g + h ( u^2 (3 - 2u) S_g + u (1 - u)^2 D_g + u^2 (u - 1) D_g+1 )
//       eps'(u) = 6 u (1 - u) S_g + (1 - u)(1 - 3u) D_g + u (3u - 2) D_g+1
//     from FOUR rows (T_g, S_g, D_g, D_g+1); the Lagrange form of rounds 1-2 read 4 + 4 rows (T and D planes), and the message
//     adjoint was bound by those L2 requests (24 of ~36 per edge).
// Against fp64 evaluation (tests/test_hip_parity.py::test_radial_filter_tables_against_float64): value 3-4e-8 of the maximum
// with either form (the fp32 rounding of T), Hermite derivative 8e-8 at FT_G = 3072 (truncation 0.008 h^3 |d4 eps/dx4| = 4e-8 +
// rounding; 1.6e-7 at 2048, 6e-8 at 4096).
struct FilterW {
  float w[4];         // Lagrange value weights at nodes g-1, g, g+1, g+2
  float a, b, c;      // Hermite value:      T_g + a S_g + b D_g + c D_g+1
  float da, db, dc;   // Hermite derivative:       da S_g + db D_g + dc D_g+1
};
__device__ __forceinline__ FilterW filter_weights(float u) {
  FilterW f;
  const float um1 = u - 1.f, um2 = u - 2.f, up1 = u + 1.f;
  f.w[0] = -u * um1 * um2 * (1.f / 6.f);
  f.w[1] = up1 * um1 * um2 * 0.5f;
  f.w[2] = -up1 * u * um2 * 0.5f;
  f.w[3] = up1 * u * um1 * (1.f / 6.f);
  const float h = 1.0f / (float)FT_G, v = 1.f - u, uu = u * u;
  f.a = h * uu * (3.f - 2.f * u);
  f.b = h * u * v * v;
  f.c = -h * uu * v;
  f.da = 6.f * u * v;
  f.db = v * (1.f - 3.f * u);
  f.dc = u * (3.f * u - 2.f);
  return f;
}
// value only: the T plane (rows g0 .. g0 + 3 = nodes g0 - 1 .. g0 + 2)
__device__ __forceinline__ float4 filter_value(const float* __restrict__ table, int g0, int c4, const FilterW& fw) {
  float4 t[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) t[k] = ld4(table + (size_t)(g0 + k) * NF + c4);
  float4 eps = mul4(t[0], fw.w[0]);
#pragma unroll
  for (int k = 1; k < 4; ++k) eps = fma4(t[k], fw.w[k], eps);
  return eps;
}
// value and derivative from four rows of the three planes
__device__ __forceinline__ void filter_value_deriv(const float* __restrict__ table, int g0, int c4, const FilterW& fw,
                                                   float4& eps, float4& deps) {
  const float* __restrict__ node = table + (size_t)(g0 + 1) * NF + c4;
  const float4 t0 = ld4(node), s0 = ld4(node + FT_PLANE), d0 = ld4(node + 2 * FT_PLANE), d1 = ld4(node + 2 * FT_PLANE + NF);
  eps = fma4(d1, fw.c, fma4(d0, fw.b, fma4(s0, fw.a, t0)));
  deps = fma4(d1, fw.dc, fma4(d0, fw.db, mul4(s0, fw.da)));
}

// The same split from the pair counts when the caller has them (pair_ptr = exclusive scan of the number of pairs a row owns, i.e. of
// its upper edges, which are the LAST ones of the row): two scalar loads next to row_ptr's instead of a vector load of the
// row's cols + ballot that the loops depend on -- one L2 round trip less at the head of every row.
__device__ __forceinline__ int row_mid(const int* __restrict__ col, int beg, int end, int i, int lane);
__device__ __forceinline__ int row_mid_of(const int* __restrict__ pair_ptr, const int* __restrict__ col, int beg, int end, int i,
                                          int lane, bool active) {
#ifndef EDGE_NO_PAIR_MID   // (tooling A/B: always take the ballot form)
  if (pair_ptr) return active ? end - (pair_ptr[i + 1] - pair_ptr[i]) : end;
#endif
  return row_mid(col, beg, end, i, lane);
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

