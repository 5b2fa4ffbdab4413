// Molecule-resident fused edge phase of an interaction layer and its adjoint (gfx950, fp32 with split-f16 products).
//
// One 8-wave workgroup per MOLECULE of at most NNHIP_MOL_STAGE_MAX atoms runs, in ONE launch per layer and direction, what the
// row path runs as three launches with two [P][128] pair arrays crossing HBM between each of them
// (newtonnet/models/newtonnet.py:207-227, the edge loop of InteractionNet.forward, and the reverse sweep torch.autograd.grad
// runs through it, newtonnet/models/output.py:66-73):
//
//   mol_edge_fwd_kernel   msg = (W_e rbf) * m[i] * m[j]            (:210-211)   formed from LDS-staged m rows, never stored
//                         a_mid = a_in + scatter_sum(msg, i)       (:213-215)   through an LDS tile, atom-owner order
//                         phi_k = W_k2 silu(W_k0 msg), k = 1, 2    (:218,:222)  MFMA, a wave owns a 32-pair tile, the hidden
//                                                                               tile never leaves its registers
//                         f_out = f_in + scatter_sum(phi1 (x) u + phi2 * f_in[j], i)   (:219-227)  through an LDS tile
//                         kept for the adjoint: silu'(h_k) and phi_k (the only pair rows that reach HBM: written once)
//   mol_edge_bwd_kernel   g_u, g_fin from gf and the kept phi rows; g_phi_k formed from LDS-staged gf / f_in rows straight into
//                         the MFMA operand (never stored); g_msg = sum_k ((g_phi_k W_k2) silu'(h_k)) W_k0 in registers;
//                         G = g_msg + g_a[i] + g_a[j] -> g_x, g_m through an LDS tile
//
// Pair rows that the row path writes and reads back per layer (x 512 B per undirected pair): msg (1 + 2), h1 | h2 (2 + 2),
// phi1 | phi2 (2 + 6), g_phi1 | g_phi2 (2 + 2), g_msg (1 + 2) = 22.  Here: silu'(h) and phi written once, read once or twice = 4 + 5.
//
// Work split inside the workgroup.  Dense stages: wave t owns pair tile t of the molecule (at most MF_ROUND_TILES tiles of 32 pairs
// per round; aspirin has 5-6), exactly the per-lane layout of mlp128s.hip -- lane (r, h) holds features 8 t + 4 h + {0..3} of pair
// row r, the stage-1 accumulators ARE the stage-2 B operand.  The weight images (f16 hi / lo planes in MFMA fragment order, made
// once per parameter set by node128s.hip:weight_image_kernel) are copied verbatim into LDS, ONE matrix at a time, and every wave
// reads its A fragments from there as contiguous 1 KiB pieces.  Aggregations: the tile owners write their rows into an LDS tile
// array [pairs][132], then wave w sums the incidences of atoms w, w + 8, w + 16 in the row path's order (even edges in the lower
// half-wave, odd ones in the upper, folded once): deterministic, no float atomics, independent of the order of the molecules.
//
// Formats: every array this pair of kernels reads or writes in global memory has the layout of the row path (a_mid, f_out,
// phi1 / phi2 [P][128], silu'(h) in the fragment order of mlp128s.hip's h_frag region indexed by GLOBAL pair tile, g_fin, g_m,
// g_x, g_u), so either direction can be swapped for the three launches it replaces (tests do exactly that).
#include <stdlib.h>

#include "common.h"

#include "edge_common.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define MF_WAVES 8
#define MF_THREADS (64 * MF_WAVES)
#define MF_ATOMS NNHIP_MOL_STAGE_MAX
#define MF_EDGES (MF_ATOMS * (MF_ATOMS - 1))
#define MF_PAIRS (MF_EDGES / 2)
#define MF_PITCH 132                                 // floats per LDS row: lanes that read DIFFERENT rows at one feature offset hit different banks
#define MF_ROUND_TILES 6
#define MF_ROUND_PAIRS (32 * MF_ROUND_TILES)
#define MF_TILE_BYTES (32 * MF_PITCH * 4)
#define MF_TILES_BYTES (MF_ROUND_TILES * MF_TILE_BYTES)   // 101 376
#define MF_NODE_BYTES (MF_ATOMS * MF_PITCH * 4)           // 12 672
#define MF_NODE3_BYTES (3 * MF_NODE_BYTES)                // 38 016
#define MF_WIMG_PLANE (NF * NF * 2)
#define MF_W_BYTES (2 * MF_WIMG_PLANE)                    // 65 536 (+ the inverse scale, read from global memory)
// lists (both kernels): geo [pairs] float4 | xg [pairs] int2 | owner edge [pairs] int | pij [pairs] u16 | inc [edges] u16 | rowb [atoms + 1] int
#define MF_LIST_BYTES (MF_PAIRS * 16 + MF_PAIRS * 8 + MF_PAIRS * 4 + MF_PAIRS * 2 + MF_EDGES * 2 + (MF_ATOMS + 1) * 4 + 12)
// forward: tiles / weight image | m | f_in | lists
#define MF_FWD_OFF_M MF_TILES_BYTES
#define MF_FWD_OFF_F (MF_FWD_OFF_M + MF_NODE_BYTES)
#define MF_FWD_OFF_LIST (MF_FWD_OFF_F + MF_NODE3_BYTES)
#define MF_FWD_LDS (MF_FWD_OFF_LIST + MF_LIST_BYTES)
// adjoint: gf | f_in | weight image, overlaid later by m | g_a | tiles; the kept phi rows are staged behind gf
#define MF_BWD_OFF_F MF_NODE3_BYTES
#define MF_BWD_OFF_W (2 * MF_NODE3_BYTES)
#define MF_BWD_OFF_PHI MF_NODE3_BYTES
#define MF_BWD_OFF_GA MF_NODE_BYTES
#define MF_BWD_OFF_TILES (2 * MF_NODE_BYTES)
#define MF_BWD_OFF_LIST (MF_BWD_OFF_W + MF_W_BYTES)
#define MF_BWD_LDS (MF_BWD_OFF_LIST + MF_LIST_BYTES)
static_assert(MF_FWD_LDS <= 163840 && MF_BWD_LDS <= 163840, "LDS budget");
static_assert(MF_BWD_OFF_PHI + MF_TILES_BYTES <= MF_BWD_OFF_LIST && MF_BWD_OFF_TILES + MF_TILES_BYTES <= MF_BWD_OFF_LIST, "overlay");
static_assert(MF_W_BYTES <= MF_TILES_BYTES, "weight image inside the tile region");

// S = 2^(14 - floor(log2 m)) as (S, 1/S); (1, 1) for zero / tiny / non-finite m   (mlp128s.hip:pow2_scale)
__device__ __forceinline__ void mf_pow2_scale(float m, float& S, float& inv) {
  const int e = (int)((__float_as_uint(m) >> 23) & 0xffu);
  const bool ok = e >= 40 && e < 255;
  S = ok ? __uint_as_float((unsigned)(268 - e) << 23) : 1.0f;
  inv = ok ? __uint_as_float((unsigned)(e - 14) << 23) : 1.0f;
}
__device__ __forceinline__ float mf_amax4(float m, const float4& v) {
  return fmaxf(fmaxf(fmaxf(m, fabsf(v.x)), fmaxf(fabsf(v.y), fabsf(v.z))), fabsf(v.w));
}
__device__ __forceinline__ void mf_split8(const float (&v)[8], float S, h8& hi, h8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float s = v[j] * S;
    const _Float16 a = (_Float16)s;
    hi[j] = a;
    lo[j] = (_Float16)(s - (float)a);
  }
}
// a lane's row (x[t] = features 8 t + 4 h + {0..3}) -> the split B operand of 8 MFMA steps; returns the inverse row scale
__device__ __forceinline__ float mf_split_row(const float4 (&x)[16], h8 (&bh)[8], h8 (&bl)[8]) {
  float m = 0.f;
#pragma unroll
  for (int t = 0; t < 16; ++t) m = mf_amax4(m, x[t]);
  m = fmaxf(m, __shfl_xor(m, 32));
  float S, inv;
  mf_pow2_scale(m, S, inv);
#pragma unroll
  for (int T = 0; T < 8; ++T) {
    const float v[8] = {x[2 * T].x, x[2 * T].y, x[2 * T].z, x[2 * T].w, x[2 * T + 1].x, x[2 * T + 1].y, x[2 * T + 1].z, x[2 * T + 1].w};
    mf_split8(v, S, bh[T], bl[T]);
  }
  return inv;
}
// the same for a stage-1 register tile hs[nb][4 q + c] = feature 32 nb + 8 q + 4 h + c (MFMA step T takes hs[T >> 1][8 (T & 1) .. + 7])
__device__ __forceinline__ float mf_split_tile(const float (&hs)[4][16], h8 (&bh)[8], h8 (&bl)[8]) {
  float m = 0.f;
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int k = 0; k < 16; ++k) m = fmaxf(m, fabsf(hs[nb][k]));
  m = fmaxf(m, __shfl_xor(m, 32));
  float S, inv;
  mf_pow2_scale(m, S, inv);
#pragma unroll
  for (int T = 0; T < 8; ++T) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = hs[T >> 1][8 * (T & 1) + j];
    mf_split8(v, S, bh[T], bl[T]);
  }
  return inv;
}
// one 32-feature block nb of D^T = W . B^T from the split operands, A fragments from the LDS copy of a fragment-order weight image
// (8 MFMA triples, fragments requested two triples ahead: mlp128s.hip:split_block with the image's addressing)
__device__ __forceinline__ f32x16 mf_block(const char* img, int nb, int lane, const h8 (&bh)[8], const h8 (&bl)[8]) {
  f32x16 acc;
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.f;
  const char* w = img + (((nb * 8) * 64 + lane) << 4);
  h8 ah0 = *reinterpret_cast<const h8*>(w), al0 = *reinterpret_cast<const h8*>(w + MF_WIMG_PLANE);
  h8 ah1 = *reinterpret_cast<const h8*>(w + 1024), al1 = *reinterpret_cast<const h8*>(w + MF_WIMG_PLANE + 1024);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int T = 0; T < 8; ++T) {
    h8 ah2, al2;
    if (T < 6) {
      ah2 = *reinterpret_cast<const h8*>(w + 1024 * (T + 2));
      al2 = *reinterpret_cast<const h8*>(w + MF_WIMG_PLANE + 1024 * (T + 2));
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
// a weight image (two f16 planes, fragment order) from the prepared block into LDS, verbatim
__device__ __forceinline__ void mf_load_w(char* dst, const char* __restrict__ img) {
  float4 v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = reinterpret_cast<const float4*>(img)[threadIdx.x + MF_THREADS * k];
#pragma unroll
  for (int k = 0; k < 8; ++k) reinterpret_cast<float4*>(dst)[threadIdx.x + MF_THREADS * k] = v[k];
}
__device__ __forceinline__ float mf_w_inv(const char* __restrict__ img) { return *reinterpret_cast<const float*>(img + MF_W_BYTES); }

// tooling (-DMF_CLOCK_DEBUG through build.sh): wall-clock stamps (100 MHz) behind the barriers of the first and the last workgroup
struct MfDbg {
#ifdef MF_CLOCK_DEBUG
  long long w[40];
  int n;
  __device__ __forceinline__ void init() { n = 0; w[n++] = wall_clock64(); }
  __device__ __forceinline__ void stamp() { if (n < 40) w[n++] = wall_clock64(); }
  __device__ __forceinline__ void print(const char* tag) {
    if ((blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) && threadIdx.x == 0) {
      printf("%s b%d:", tag, (int)blockIdx.x);
      for (int k = 1; k < n; ++k) printf(" %.2f", (double)(w[k] - w[k - 1]) / 100.0);
      printf("  total %.2f us\n", (double)(w[n - 1] - w[0]) / 100.0);
    }
  }
#else
  __device__ __forceinline__ void init() {}
  __device__ __forceinline__ void stamp() {}
  __device__ __forceinline__ void print(const char*) {}
#endif
};

struct MfLists {
  float4* geo;            // [pairs] (u, r) of the owner's direction i -> j, i < j
  int2* xg;               // [pairs] position on the radial-filter grid
  int* pe;                // [pairs] global index of the owner's directed edge
  unsigned short* pij;    // [pairs] i_local | j_local << 8
  unsigned short* inc;    // [edges] pair_local | other_local << 9 | (this row owns the pair) << 14, in row order
  int* rowb;              // [atoms + 1] local edge offsets
};
__device__ __forceinline__ MfLists mf_lists(char* base) {
  MfLists L;
  L.geo = reinterpret_cast<float4*>(base);
  L.xg = reinterpret_cast<int2*>(base + MF_PAIRS * 16);
  L.pe = reinterpret_cast<int*>(base + MF_PAIRS * 24);
  L.pij = reinterpret_cast<unsigned short*>(base + MF_PAIRS * 28);
  L.inc = reinterpret_cast<unsigned short*>(base + MF_PAIRS * 30);
  L.rowb = reinterpret_cast<int*>(base + MF_PAIRS * 30 + MF_EDGES * 2);
  return L;
}
// the molecule's extents; false: nothing to do here (an empty slot, or a molecule this form does not serve -- the caller launches it
// only for batches whose status word had bit 8 clear, and a deferred step whose guess was wrong is repeated by the host)
struct MfMol {
  int a0, n, E0, nE, P0, nP;
};
__device__ __forceinline__ bool mf_molecule(const int* __restrict__ mol_ptr, const int* __restrict__ row_ptr,
                                            const int* __restrict__ pair_ptr, int b, MfMol& M) {
  M.a0 = mol_ptr[b];
  M.n = mol_ptr[b + 1] - M.a0;
  if (M.n <= 0 || M.n > MF_ATOMS) return false;
  M.E0 = row_ptr[M.a0];
  M.nE = row_ptr[M.a0 + M.n] - M.E0;
  M.P0 = pair_ptr[M.a0];
  M.nP = pair_ptr[M.a0 + M.n] - M.P0;
  return M.nE >= 0 && M.nE <= MF_EDGES && M.nP >= 0 && M.nP <= MF_PAIRS && M.nE == 2 * M.nP;
}
// (call with rowb already visible) one thread per directed edge of the molecule
__device__ __forceinline__ void mf_build_lists(const MfLists& L, const MfMol& M, const int* __restrict__ col, const int* __restrict__ pid,
                                               const float* __restrict__ geo, const int2* __restrict__ xg) {
  for (int el = threadIdx.x; el < M.nE; el += MF_THREADS) {
    int k = 0;
    while (k + 1 < M.n && L.rowb[k + 1] <= el) ++k;
    const int e = M.E0 + el;
    int j = col[e] - M.a0, p = pid[e] - M.P0;
    j = min(max(j, 0), M.n - 1);             // (a valid list never needs these; keeps every LDS index inside its array)
    p = min(max(p, 0), max(M.nP - 1, 0));
    const bool own = j > k;
    L.inc[el] = (unsigned short)(p | (j << 9) | (own ? (1 << 14) : 0));
    if (own) {
      L.pij[p] = (unsigned short)(k | (j << 8));
      L.geo[p] = reinterpret_cast<const float4*>(geo)[e];
      L.xg[p] = xg[e];
      L.pe[p] = e;
    }
  }
}
// the part [eb, ee) of atom a's incidence list that lies in the round of pairs [pb, pb + MF_ROUND_PAIRS) (pairs ascend along a row)
__device__ __forceinline__ void mf_row_range(const MfLists& L, int a, int pb, int nP, int lane, int& eb, int& ee) {
  const int beg = L.rowb[a], end = L.rowb[a + 1];
  eb = beg;
  ee = end;
  if (nP > MF_ROUND_PAIRS) {   // (uniform; a row has at most 23 < 64 edges)
    const int p = beg + lane < end ? (int)(L.inc[beg + lane] & 511) : 0x7fff;
    eb = beg + __popcll(__ballot(p < pb));
    ee = beg + __popcll(__ballot(p < pb + MF_ROUND_PAIRS));
  }
}
// rows [row0, row0 + n_rows) of a [.][128] array -> LDS rows of MF_PITCH floats
__device__ __forceinline__ void mf_stage_rows(float* dst, const float* __restrict__ src, size_t row0, int n_rows) {
  const float4* s = reinterpret_cast<const float4*>(src + row0 * NF);
  for (int t = threadIdx.x; t < n_rows * 32; t += MF_THREADS)
    *reinterpret_cast<float4*>(dst + (t >> 5) * MF_PITCH + 4 * (t & 31)) = s[t];
}
__device__ __forceinline__ float4 lds4(const float* p) { return *reinterpret_cast<const float4*>(p); }

struct MolFwdArgs {
  const int *mol_ptr, *row_ptr, *pair_ptr, *col, *pid;
  const float* geo;
  const int2* xg;
  const float *m, *a_in, *f_in, *table;
  const char *img10, *img12, *img20, *img22;
  float *a_mid, *f_out, *h1, *h2, *phi1, *phi2;
  int n_mol;
};

// One fused Linear-SiLU-Linear over the tiles of the round, stage by stage with one weight image in LDS at a time.
//   FWD:  hs = silu(X W1^T) with silu'(.) kept in Hk (global-tile fragment order);  y = hs W2^T
//   BWD:  hs = (X W1^T) * Hk;                                                       y (+)= hs W2^T     (ACC: y holds the other MLP's term)
// Every wave must call it (workgroup barriers inside).  On entry nobody may still be using the tile region; on exit it holds W2.
template <bool FWD, bool ACC>
__device__ __forceinline__ void mf_mlp(char* wlds, const char* __restrict__ img1, const char* __restrict__ img2, const h8 (&xh)[8],
                                       const h8 (&xl)[8], float invx, float* __restrict__ Hk, size_t pg, bool tile_on, bool live,
                                       int lane, float (&y)[4][16], MfDbg& dbg) {
  const int h = lane >> 5;
  const size_t tile_g = pg >> 5;
  const int lane_g = 32 * h + (int)(pg & 31);
  mf_load_w(wlds, img1);
  __syncthreads();
  dbg.stamp();
  float hs[4][16];
  if (tile_on) {
    const float inv1 = invx * mf_w_inv(img1);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      float4* hp = reinterpret_cast<float4*>(Hk) + (tile_g * 4 + nb) * 256 + lane_g;
      float4 hin[4];
      if (!FWD) {
#pragma unroll
        for (int q = 0; q < 4; ++q) hin[q] = live ? ld4_nt(reinterpret_cast<const float*>(hp + 64 * q)) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      const f32x16 acc = mf_block(wlds, nb, lane, xh, xl);
      if (FWD) {
        float keep[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const float v = acc[k] * inv1;
          const float s = sigmoid_f(v);
          keep[k] = s * (1.0f + v * (1.0f - s));    // silu'(h): all the adjoint wants of h (shares the sigmoid with the activation)
          hs[nb][k] = v * s;
        }
        if (live) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            st4_nt(reinterpret_cast<float*>(hp + 64 * q), make_float4(keep[4 * q], keep[4 * q + 1], keep[4 * q + 2], keep[4 * q + 3]));
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          hs[nb][4 * q] = acc[4 * q] * inv1 * hin[q].x;
          hs[nb][4 * q + 1] = acc[4 * q + 1] * inv1 * hin[q].y;
          hs[nb][4 * q + 2] = acc[4 * q + 2] * inv1 * hin[q].z;
          hs[nb][4 * q + 3] = acc[4 * q + 3] * inv1 * hin[q].w;
        }
      }
    }
  }
  __syncthreads();             // every wave is done with W1
  dbg.stamp();
  mf_load_w(wlds, img2);
  __syncthreads();
  dbg.stamp();
  if (tile_on) {
    h8 bh[8], bl[8];
    const float inv2 = mf_split_tile(hs, bh, bl) * mf_w_inv(img2);
#pragma unroll
    for (int nb2 = 0; nb2 < 4; ++nb2) {
      const f32x16 acc = mf_block(wlds, nb2, lane, bh, bl);
#pragma unroll
      for (int k = 0; k < 16; ++k) y[nb2][k] = ACC ? fmaf(acc[k], inv2, y[nb2][k]) : acc[k] * inv2;
    }
  }
}
// this lane's stage-2 tile y[nb][4 q + c] = feature 32 nb + 8 q + 4 h + c of row `row` -> an LDS tile row / a [.][128] global row
__device__ __forceinline__ void mf_tile_put(float* tiles, int row, int h, const float (&y)[4][16]) {
  float* p = tiles + row * MF_PITCH + 4 * h;
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<float4*>(p + 32 * nb + 8 * q) = make_float4(y[nb][4 * q], y[nb][4 * q + 1], y[nb][4 * q + 2], y[nb][4 * q + 3]);
}
__device__ __forceinline__ void mf_row_put(float* __restrict__ Y, size_t pg, int h, const float (&y)[4][16]) {
  float* p = Y + pg * NF + 4 * h;
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) st4(p + 32 * nb + 8 * q, make_float4(y[nb][4 * q], y[nb][4 * q + 1], y[nb][4 * q + 2], y[nb][4 * q + 3]));
}

// -----------------------------------------------------------------------------------------------------------------------
// forward
// -----------------------------------------------------------------------------------------------------------------------
template <bool HAS_F>
__global__ void __launch_bounds__(MF_THREADS, 2) mol_edge_fwd_kernel(const MolFwdArgs A) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float* tiles = reinterpret_cast<float*>(lds);
  float* sm_m = reinterpret_cast<float*>(lds + MF_FWD_OFF_M);
  float* sm_f = reinterpret_cast<float*>(lds + MF_FWD_OFF_F);
  const MfLists L = mf_lists(lds + MF_FWD_OFF_LIST);

  const int b = blockIdx.x;
  MfMol M;
  if (b >= A.n_mol || !mf_molecule(A.mol_ptr, A.row_ptr, A.pair_ptr, b, M)) return;   // (uniform)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5, c4 = 4 * r;
  const bool hi = h != 0;
  const int a0 = M.a0, n = M.n, nP = M.nP;
  MfDbg dbg;
  dbg.init();

  if (tid <= n) L.rowb[tid] = A.row_ptr[a0 + tid] - M.E0;
  mf_stage_rows(sm_m, A.m, (size_t)a0, n);
  if (HAS_F) mf_stage_rows(sm_f, A.f_in, (size_t)a0 * 3, 3 * n);
  __syncthreads();
    dbg.stamp();
  mf_build_lists(L, M, A.col, A.pid, A.geo, A.xg);
  __syncthreads();
    dbg.stamp();

  for (int pb = 0; pb == 0 || pb < nP; pb += MF_ROUND_PAIRS) {
    const int nPr = min(nP - pb, MF_ROUND_PAIRS);        // pairs of this round (0 for a molecule without edges)
    const int nT = (nPr + 31) >> 5;
    // ---- radial filter rows eps[p] (message_edgepart, newtonnet.py:186,210): a half-wave per pair, two pairs in flight
    for (int p = 2 * wave + h; p < nPr; p += 4 * MF_WAVES) {
      const int q = p + 2 * MF_WAVES;
      const int2 g0 = L.xg[pb + p], g1 = L.xg[pb + min(q, nPr - 1)];
      const FilterW w0 = filter_weights(__int_as_float(g0.y)), w1 = filter_weights(__int_as_float(g1.y));
      const float4 e0 = filter_value(A.table, g0.x, c4, w0);
      const float4 e1 = filter_value(A.table, g1.x, c4, w1);
      *reinterpret_cast<float4*>(tiles + p * MF_PITCH + c4) = e0;
      if (q < nPr) *reinterpret_cast<float4*>(tiles + q * MF_PITCH + c4) = e1;
    }
    __syncthreads();
    dbg.stamp();
    // ---- msg tiles: lane (r, h) of wave t forms its half of pair row 32 t + r in the MFMA operand layout, leaves the row in the
    // tile for the invariant aggregation and keeps the split operand for both MLPs
    const bool tile_on = wave < nT;
    const int pl = 32 * wave + r;
    const bool live = tile_on && pl < nPr;
    const size_t pg = (size_t)M.P0 + pb + pl;            // global pair row (meaningful when live)
    h8 xh[8], xl[8];
    float invx = 1.0f;
    if (tile_on) {
      float4 x[16];
      const int ij = live ? (int)L.pij[pb + pl] : 0;
      const float* mi = sm_m + (ij & 255) * MF_PITCH + 4 * h;
      const float* mj = sm_m + (ij >> 8) * MF_PITCH + 4 * h;
      float* er = tiles + pl * MF_PITCH + 4 * h;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const float4 v = mul4(mul4(lds4(er + 8 * t), lds4(mi + 8 * t)), lds4(mj + 8 * t));
        x[t] = live ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) *reinterpret_cast<float4*>(er + 8 * t) = v;
      }
      invx = mf_split_row(x, xh, xl);
    }
    __syncthreads();
    dbg.stamp();
    // ---- a_mid[a] = a_in[a] + sum of the messages of a's edges (newtonnet.py:213-215)
    for (int a = wave; a < n; a += MF_WAVES) {
      int eb, ee;
      mf_row_range(L, a, pb, nP, lane, eb, ee);
      const float* base = (pb == 0 ? A.a_in : A.a_mid) + (size_t)(a0 + a) * NF + c4;
      const float4 b0 = hi ? make_float4(0.f, 0.f, 0.f, 0.f) : ld4(base);
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int e = eb + h; e < ee; e += 2) acc = add4(acc, lds4(tiles + ((int)(L.inc[e] & 511) - pb) * MF_PITCH + c4));
      acc = add4(acc, upper_half(acc));
      if (!hi) st4(A.a_mid + (size_t)(a0 + a) * NF + c4, add4(b0, acc));
    }
    __syncthreads();                                      // the tile region is free for the weights
    dbg.stamp();
    // ---- equiv_message1 (newtonnet.py:218) and the phi1 (x) u half of the force messages (:219-220)
    float y[4][16];
    mf_mlp<true, false>(lds, A.img10, A.img12, xh, xl, invx, A.h1, pg, tile_on, live, lane, y, dbg);
    if (live) mf_row_put(A.phi1, pg, h, y);
    __syncthreads();                                      // every wave is done with W2
    dbg.stamp();
    if (tile_on) mf_tile_put(tiles, pl, h, y);
    __syncthreads();
    dbg.stamp();
    for (int a = wave; a < n; a += MF_WAVES) {
      int eb, ee;
      mf_row_range(L, a, pb, nP, lane, eb, ee);
      float4 acc[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int e = eb + h; e < ee; e += 2) {
        const int inc = L.inc[e], p = inc & 511;
        const float4 g = L.geo[p];
        const float s = (inc >> 14) & 1 ? 1.0f : -1.0f;   // u of the reverse direction is -u
        const float4 v1 = lds4(tiles + (p - pb) * MF_PITCH + c4);
        acc[0] = fma4(v1, s * g.x, acc[0]);
        acc[1] = fma4(v1, s * g.y, acc[1]);
        acc[2] = fma4(v1, s * g.z, acc[2]);
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float4 o = add4(acc[k], upper_half(acc[k]));
        if (!hi) {
          float4 base = make_float4(0.f, 0.f, 0.f, 0.f);
          if (pb > 0)
            base = ld4(A.f_out + ((size_t)(a0 + a) * 3 + k) * NF + c4);
          else if (HAS_F)
            base = lds4(sm_f + (a * 3 + k) * MF_PITCH + c4);
          st4(A.f_out + ((size_t)(a0 + a) * 3 + k) * NF + c4, add4(base, o));
        }
      }
    }
    if (HAS_F) {
      __syncthreads();
    dbg.stamp();
      // ---- equiv_message2 (newtonnet.py:222) and the phi2 * force_node[j] half (:223-224)
      mf_mlp<true, false>(lds, A.img20, A.img22, xh, xl, invx, A.h2, pg, tile_on, live, lane, y, dbg);
      if (live) mf_row_put(A.phi2, pg, h, y);
      __syncthreads();
    dbg.stamp();
      if (tile_on) mf_tile_put(tiles, pl, h, y);
      __syncthreads();
    dbg.stamp();
      for (int a = wave; a < n; a += MF_WAVES) {
        int eb, ee;
        mf_row_range(L, a, pb, nP, lane, eb, ee);
        float4 base[3], acc[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {   // (what this very thread wrote a moment ago: requested ahead of the loop)
          base[k] = hi ? make_float4(0.f, 0.f, 0.f, 0.f) : ld4(A.f_out + ((size_t)(a0 + a) * 3 + k) * NF + c4);
          acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        for (int e = eb + h; e < ee; e += 2) {
          const int inc = L.inc[e], p = inc & 511, j = (inc >> 9) & 31;
          const float4 v2 = lds4(tiles + (p - pb) * MF_PITCH + c4);
#pragma unroll
          for (int k = 0; k < 3; ++k) acc[k] = fma4(v2, lds4(sm_f + (j * 3 + k) * MF_PITCH + c4), acc[k]);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float4 o = add4(acc[k], upper_half(acc[k]));
          if (!hi) st4(A.f_out + ((size_t)(a0 + a) * 3 + k) * NF + c4, add4(base[k], o));
        }
      }
    }
    __syncthreads();                                      // (the next round reuses the tile region)
    dbg.stamp();
  }
  dbg.print(HAS_F ? "mol_fwd<1>" : "mol_fwd<0>");
}

// -----------------------------------------------------------------------------------------------------------------------
// adjoint
// -----------------------------------------------------------------------------------------------------------------------
struct MolBwdArgs {
  const int *mol_ptr, *row_ptr, *pair_ptr, *col, *pid, *rev;
  const float* geo;
  const int2* xg;
  const float *gf, *g_a, *m, *f_in, *table;
  const char *img12T, *img10T, *img22T, *img20T;
  const float *h1, *h2, *phi1, *phi2;
  float *g_fin, *g_m, *g_x, *g_u;
  int n_mol;
};

// LOWER = false for the first layer: force_node == 0 entering it (no phi2 branch, no g_fin) and its m does not depend on the
// positions (no g_m) -- edge.hip: force_bwd_kernel<false>, msg_bwd_kernel<false>.
template <bool LOWER>
__global__ void __launch_bounds__(MF_THREADS, 2) mol_edge_bwd_kernel(const MolBwdArgs A) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float* sm_gf = reinterpret_cast<float*>(lds);
  float* sm_f = reinterpret_cast<float*>(lds + MF_BWD_OFF_F);
  float* phi_t = reinterpret_cast<float*>(lds + MF_BWD_OFF_PHI);       // the kept phi rows of the round (behind gf, over f_in | W)
  char* wlds = lds + MF_BWD_OFF_W;
  float* sm_m = reinterpret_cast<float*>(lds);                          // overlay once gf / f_in / W are dead
  float* sm_ga = reinterpret_cast<float*>(lds + MF_BWD_OFF_GA);
  float* tiles = reinterpret_cast<float*>(lds + MF_BWD_OFF_TILES);
  const MfLists L = mf_lists(lds + MF_BWD_OFF_LIST);

  const int b = blockIdx.x;
  MfMol M;
  if (b >= A.n_mol || !mf_molecule(A.mol_ptr, A.row_ptr, A.pair_ptr, b, M)) return;   // (uniform)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5, c4 = 4 * r;
  const bool hi = h != 0;
  const int a0 = M.a0, n = M.n, nP = M.nP;
  MfDbg dbg;
  dbg.init();

  if (tid <= n) L.rowb[tid] = A.row_ptr[a0 + tid] - M.E0;
  __syncthreads();
    dbg.stamp();
  mf_build_lists(L, M, A.col, A.pid, A.geo, A.xg);

  for (int pb = 0; pb == 0 || pb < nP; pb += MF_ROUND_PAIRS) {
    const int nPr = min(nP - pb, MF_ROUND_PAIRS);
    const int nT = (nPr + 31) >> 5;
    const size_t pg0 = (size_t)M.P0 + pb;
    // ---- stage gf and the kept phi1 rows of the round
    mf_stage_rows(sm_gf, A.gf, (size_t)a0 * 3, 3 * n);
    mf_stage_rows(phi_t, A.phi1, pg0, nPr);
    __syncthreads();
    dbg.stamp();
    // ---- g_u of both directions of every pair: < gf[i][k], phi1[p] > and < gf[j][k], phi1[p] >   (edge.hip:force_bwd_kernel)
    for (int p = 2 * wave + h; p < nPr; p += 2 * MF_WAVES) {
      const int ij = L.pij[pb + p], i = ij & 255, j = ij >> 8;
      const float4 v1 = lds4(phi_t + p * MF_PITCH + c4);
      float si[3], sj[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        si[k] = half_sum_top(dot4(lds4(sm_gf + (i * 3 + k) * MF_PITCH + c4), v1));
        sj[k] = half_sum_top(dot4(lds4(sm_gf + (j * 3 + k) * MF_PITCH + c4), v1));
      }
      if (r == 31) {
        const int e = L.pe[pb + p];
        reinterpret_cast<float4*>(A.g_u)[e] = make_float4(si[0], si[1], si[2], 0.f);
        reinterpret_cast<float4*>(A.g_u)[A.rev[e]] = make_float4(sj[0], sj[1], sj[2], 0.f);
      }
    }
    if (LOWER) {
      __syncthreads();
    dbg.stamp();
      mf_stage_rows(phi_t, A.phi2, pg0, nPr);
      __syncthreads();
    dbg.stamp();
      // ---- g_fin[a][k] = gf[a][k] + sum_{e in row a} phi2[p] * gf[j][k]
      for (int a = wave; a < n; a += MF_WAVES) {
        int eb, ee;
        mf_row_range(L, a, pb, nP, lane, eb, ee);
        float4 base[3], acc[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          base[k] = hi ? make_float4(0.f, 0.f, 0.f, 0.f)
                       : (pb == 0 ? lds4(sm_gf + (a * 3 + k) * MF_PITCH + c4) : ld4(A.g_fin + ((size_t)(a0 + a) * 3 + k) * NF + c4));
          acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        for (int e = eb + h; e < ee; e += 2) {
          const int inc = L.inc[e], p = inc & 511, j = (inc >> 9) & 31;
          const float4 v2 = lds4(phi_t + (p - pb) * MF_PITCH + c4);
#pragma unroll
          for (int k = 0; k < 3; ++k) acc[k] = fma4(v2, lds4(sm_gf + (j * 3 + k) * MF_PITCH + c4), acc[k]);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float4 o = add4(acc[k], upper_half(acc[k]));
          if (!hi) st4(A.g_fin + ((size_t)(a0 + a) * 3 + k) * NF + c4, add4(base[k], o));
        }
      }
    }
    __syncthreads();                                      // the staged phi rows are dead: f_in and the weights move in
    dbg.stamp();
    if (LOWER) mf_stage_rows(sm_f, A.f_in, (size_t)a0 * 3, 3 * n);
    // ---- g_phi1[p] = sum_k (gf[i][k] - gf[j][k]) u_p[k], formed in the MFMA operand layout, then the adjoint of equiv_message1
    const bool tile_on = wave < nT;
    const int pl = 32 * wave + r;
    const bool live = tile_on && pl < nPr;
    const size_t pg = pg0 + pl;
    const int ij = live ? (int)L.pij[pb + pl] : 0;
    const float* gi = sm_gf + (ij & 255) * 3 * MF_PITCH + 4 * h;
    const float* gj = sm_gf + (ij >> 8) * 3 * MF_PITCH + 4 * h;
    float y[4][16];
    {
      h8 xh[8], xl[8];
      float invx = 1.0f;
      if (tile_on) {
        const float4 g = live ? L.geo[pb + pl] : make_float4(0.f, 0.f, 0.f, 1.f);
        float4 x[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          float4 v = mul4(sub4(lds4(gi + 8 * t), lds4(gj + 8 * t)), g.x);
          v = fma4(sub4(lds4(gi + MF_PITCH + 8 * t), lds4(gj + MF_PITCH + 8 * t)), g.y, v);
          v = fma4(sub4(lds4(gi + 2 * MF_PITCH + 8 * t), lds4(gj + 2 * MF_PITCH + 8 * t)), g.z, v);
          x[t] = live ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        invx = mf_split_row(x, xh, xl);
      }
      mf_mlp<false, false>(wlds, A.img12T, A.img10T, xh, xl, invx, const_cast<float*>(A.h1), pg, tile_on, live, lane, y, dbg);
    }
    if (LOWER) {
      // ---- g_phi2[p] = sum_k gf[i][k] * f_in[j][k] + gf[j][k] * f_in[i][k], then the adjoint of equiv_message2 on top
      h8 xh[8], xl[8];
      float invx = 1.0f;
      if (tile_on) {
        const float* fi = sm_f + (ij & 255) * 3 * MF_PITCH + 4 * h;
        const float* fj = sm_f + (ij >> 8) * 3 * MF_PITCH + 4 * h;
        float4 x[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            v = fma4(lds4(gi + k * MF_PITCH + 8 * t), lds4(fj + k * MF_PITCH + 8 * t), v);
            v = fma4(lds4(gj + k * MF_PITCH + 8 * t), lds4(fi + k * MF_PITCH + 8 * t), v);
          }
          x[t] = live ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        invx = mf_split_row(x, xh, xl);
      }
      __syncthreads();                                    // every wave is done with the first MLP's second matrix
    dbg.stamp();
      mf_mlp<false, true>(wlds, A.img22T, A.img20T, xh, xl, invx, const_cast<float*>(A.h2), pg, tile_on, live, lane, y, dbg);
    }
    __syncthreads();                                      // gf, f_in and the weights are dead
    dbg.stamp();
    // ---- the message adjoint (edge.hip:msg_bwd_kernel): G = g_msg[p] + g_a[i] + g_a[j]
    mf_stage_rows(sm_m, A.m, (size_t)a0, n);
    mf_stage_rows(sm_ga, A.g_a, (size_t)a0, n);
    if (tile_on) mf_tile_put(tiles, pl, h, y);
    __syncthreads();
    dbg.stamp();
    for (int p = 2 * wave + h; p < nPr; p += 2 * MF_WAVES) {
      const int ij2 = L.pij[pb + p], i = ij2 & 255, j = ij2 >> 8;
      const int2 gx = L.xg[pb + p];
      const FilterW fw = filter_weights(__int_as_float(gx.y));
      float4 eps, deps;
      filter_value_deriv(A.table, gx.x, c4, fw, eps, deps);
      float* row = tiles + p * MF_PITCH + c4;
      const float4 G = add4(add4(lds4(row), lds4(sm_ga + i * MF_PITCH + c4)), lds4(sm_ga + j * MF_PITCH + c4));
      const float4 mi = lds4(sm_m + i * MF_PITCH + c4), mj = lds4(sm_m + j * MF_PITCH + c4);
      const float gxs = half_sum_top(dot4(mul4(mul4(G, mi), mj), deps));
      if (r == 31) {       // x is shared by the two directions and only their sum enters the force: the owner's edge carries it
        const int e = L.pe[pb + p];
        A.g_x[e] = gxs;
        A.g_x[A.rev[e]] = 0.f;
      }
      if (LOWER) *reinterpret_cast<float4*>(row) = mul4(G, eps);
    }
    if (LOWER) {
      __syncthreads();
    dbg.stamp();
      // ---- g_m[a] = sum_{e in row a} (G eps)[p] * m[j]
      for (int a = wave; a < n; a += MF_WAVES) {
        int eb, ee;
        mf_row_range(L, a, pb, nP, lane, eb, ee);
        const float4 base = (hi || pb == 0) ? make_float4(0.f, 0.f, 0.f, 0.f) : ld4(A.g_m + (size_t)(a0 + a) * NF + c4);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int e = eb + h; e < ee; e += 2) {
          const int inc = L.inc[e], p = inc & 511, j = (inc >> 9) & 31;
          acc = fma4(lds4(tiles + (p - pb) * MF_PITCH + c4), lds4(sm_m + j * MF_PITCH + c4), acc);
        }
        acc = add4(acc, upper_half(acc));
        if (!hi) st4(A.g_m + (size_t)(a0 + a) * NF + c4, add4(base, acc));
      }
    }
    __syncthreads();                                      // (the next round restages gf over m / g_a / the tiles)
    dbg.stamp();
  }
  dbg.print(LOWER ? "mol_bwd<1>" : "mol_bwd<0>");
}

// -----------------------------------------------------------------------------------------------------------------------
// launchers (pipeline.hip)
// -----------------------------------------------------------------------------------------------------------------------
template <typename K>
static int mf_set_lds(K kernel, size_t bytes) {
  hipError_t rc = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (rc != hipSuccess) {
    nnhip_set_error("molfuse: hipFuncSetAttribute(%zu bytes of LDS) -> %s", bytes, hipGetErrorString(rc));
    return NNHIP_E_HIP;
  }
  return 0;
}

int launch_mol_edge_fwd(bool has_f, const int* mol_ptr, const int* row_ptr, const int* pair_ptr, const int* col, const int* pid,
                        const float* geo, const int* xg, const float* m, const float* a_in, const float* f_in, const float* table,
                        const char* img10, const char* img12, const char* img20, const char* img22, float* a_mid, float* f_out,
                        float* h1, float* h2, float* phi1, float* phi2, int n_mol, hipStream_t s) {
  ScopedTimer t0(TC_MOL_FWD, s);
  static const int rc0 = mf_set_lds(mol_edge_fwd_kernel<false>, MF_FWD_LDS), rc1 = mf_set_lds(mol_edge_fwd_kernel<true>, MF_FWD_LDS);
  if (rc0 || rc1) return NNHIP_E_HIP;
  if (n_mol <= 0) return 0;
  MolFwdArgs A = {mol_ptr, row_ptr, pair_ptr, col, pid, geo, reinterpret_cast<const int2*>(xg), m, a_in, f_in, table,
                  img10, img12, img20, img22, a_mid, f_out, h1, h2, phi1, phi2, n_mol};
  if (has_f)
    mol_edge_fwd_kernel<true><<<n_mol, MF_THREADS, MF_FWD_LDS, s>>>(A);
  else
    mol_edge_fwd_kernel<false><<<n_mol, MF_THREADS, MF_FWD_LDS, s>>>(A);
  LAUNCH_CHECK();
  return 0;
}

int launch_mol_edge_bwd(bool lower, const int* mol_ptr, const int* row_ptr, const int* pair_ptr, const int* col, const int* pid,
                        const int* rev, const float* geo, const int* xg, const float* gf, const float* g_a, const float* m,
                        const float* f_in, const float* table, const char* img12T, const char* img10T, const char* img22T,
                        const char* img20T, const float* h1, const float* h2, const float* phi1, const float* phi2, float* g_fin,
                        float* g_m, float* g_x, float* g_u, int n_mol, hipStream_t s) {
  ScopedTimer t0(TC_MOL_BWD, s);
  static const int rc0 = mf_set_lds(mol_edge_bwd_kernel<false>, MF_BWD_LDS), rc1 = mf_set_lds(mol_edge_bwd_kernel<true>, MF_BWD_LDS);
  if (rc0 || rc1) return NNHIP_E_HIP;
  if (n_mol <= 0) return 0;
  MolBwdArgs A = {mol_ptr, row_ptr, pair_ptr, col, pid, rev, geo, reinterpret_cast<const int2*>(xg), gf, g_a, m, f_in, table,
                  img12T, img10T, img22T, img20T, h1, h2, phi1, phi2, g_fin, g_m, g_x, g_u, n_mol};
  if (lower)
    mol_edge_bwd_kernel<true><<<n_mol, MF_THREADS, MF_BWD_LDS, s>>>(A);
  else
    mol_edge_bwd_kernel<false><<<n_mol, MF_THREADS, MF_BWD_LDS, s>>>(A);
  LAUNCH_CHECK();
  return 0;
}
