// Molecule-resident fused edge phase: messages, both edge MLPs, the force-message sums (and their adjoints) of a layer in ONE launch,
// pair rows on chip.  TWO 4-wave workgroups per CU (256 threads x 256 registers, 80 KB of LDS each), so that two molecules share a CU
// and one's latencies run under the other's matrix work.  (Round 5's first form -- one 8-wave workgroup per CU with a weight matrix in
// LDS, 158 KB -- lost at every batch size and was removed in round 6; profiles/HISTORY_r05.md has its numbers.)
//
// SCHEDULE (round 6): a PERSISTENT grid of at most 2 x 256 workgroups.  A workgroup takes molecules from a device-side queue -- one
// returning atomicAdd on a head word per molecule, issued a whole molecule ahead -- over an ORDER that mol2_order_kernel builds once
// per step from the finished neighbor list: molecules by their number of 32-pair tiles, largest first (a stable counting sort), so a
// launch ends within one SMALL molecule of balanced instead of at the boundary of a round of workgroups that lasts as long as its
// largest molecule (round 5: a mix of the nine MD17 shapes lost 33-53 % to the row path, 640 conformers cost two rounds).  Results do
// not depend on the schedule: a molecule's arithmetic is its own.  Layer 0 (one edge MLP) loads its weight fragments once per launch.
// What makes two workgroups per CU fit:
//   * the edge MLPs stream the molecule's pair tiles ONE AT A TIME through a 17 KB operand tile (node128s.hip's row-local scheme:
//     wave w = output block w of every GEMM, the weight fragments of its block resident in REGISTERS for all tiles of the molecule --
//     2 x 64 registers per MLP, read once per workgroup from the fragment-order images; rows scaled by their maximum through a
//     publish / commit exchange; the hidden tile and the phi tile share one LDS buffer);
//   * the messages are formed row by row as msg_fwd_kernel does (m staged in LDS; msg rows go to the L2 and come back tile by tile
//     microseconds later), so neither the radial-filter rows nor all message tiles ever sit in LDS together;
//   * the force-message sums live in 36 registers per lane (wave w owns atoms w, w + 4, ...; a full wave per incidence, two features
//     per lane), added tile by tile in pair order: deterministic, no float atomics.
// Global formats are the row path's (a_mid, f_out, msg, phi1 / phi2 [P][128], silu'(h) in mlp128s.hip's fragment order by GLOBAL
// pair tile): the adjoint of either path can follow.
// Reference semantics: newtonnet/models/newtonnet.py:207-227.
//
// BUILD NOTE: compiled WITHOUT packed-fp32 instructions, like every file of the library (build.sh).  This is the kernel in which the
// problem behind that switch was found: with v_pk_fma_f32 / v_pk_mul_f32 chains for its float4 arithmetic, about one step in 200 of
// 1024 molecules returned ONE molecule slightly wrong -- a radial-filter value short of exactly one of its four interpolation terms,
// in the low half of one packed register pair x 16 lanes.  profiles/r05_mol_fused2_soak.txt has the hunt, tools/probes/
// pk_chain_probe.hip the stand-alone reproducer (code-alignment dependent, two or more waves per SIMD, v_fma_f32 never fails).
#include <stdlib.h>

#include "common.h"

#include "edge_common.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

#define M2_WAVES 4
#define M2_THREADS (64 * M2_WAVES)
#define M2_GRID 512                                   // persistent workgroups: two per CU (80 KB of LDS each)
#define M2_ATOMS NNHIP_MOL_STAGE_MAX
#define M2_EDGES (M2_ATOMS * (M2_ATOMS - 1))
#define M2_PAIRS (M2_EDGES / 2)
#define M2_MAX_TILES ((M2_PAIRS + 31) / 32)           // 9
#define M2_OWN ((M2_ATOMS + M2_WAVES - 1) / M2_WAVES) // atoms per wave: 6
#define M2_PITCH 272                                  // bytes per row of one f16 plane (128 f16 + 16 pad: conflict-free ds_read_b128)
#define M2_PLANE (32 * M2_PITCH)
#define M2_TILE (2 * M2_PLANE)                        // 17 408
#ifndef M2_KNOWN_MAX
#define M2_KNOWN_MAX 1                                // the forward's operand rows carry their maximum from the message pass (no exchange)
#endif
#ifndef M2_AGG
#define M2_AGG 2                                      // incidences an owner wave keeps in flight per LDS round trip (4 is no faster and raises
                                                      // the rate of the intermittent error described in the header ~10x)
#endif
#define M2_PHI_PITCH 132                              // floats per row of the fp32 phi tile (aliases the hidden tile)
#define M2_WIMG_PLANE (NF * NF * 2)
// LDS: operand tile | hidden tile (= phi tile) | row maxima | f_in | geo | xg | (spare) | rowb | pij;  m aliases the operand tile in pass 1
#define M2_OFF_H M2_TILE
#define M2_OFF_PMAX (2 * M2_TILE)
#define M2_OFF_F (M2_OFF_PMAX + 8 * 32 * 4)
#define M2_OFF_GEO (M2_OFF_F + 3 * M2_ATOMS * NF * 4)
#define M2_OFF_XG (M2_OFF_GEO + M2_PAIRS * 16)
#define M2_OFF_INC (M2_OFF_XG + M2_PAIRS * 8)
#define M2_OFF_ROWB (M2_OFF_INC + M2_EDGES * 2)
#define M2_OFF_PIJ (M2_OFF_ROWB + (M2_ATOMS + 1) * 4 + 4)
#define M2_OFF_Q (M2_OFF_PIJ + M2_PAIRS * 2 + 8)        // two queue slots (the next molecule's position, double-buffered)
#define M2_LDS (M2_OFF_Q + 8)
static_assert(M2_LDS <= 81920, "two workgroups per CU");
static_assert(M2_OFF_Q % 4 == 0, "queue slots are ints");
static_assert(32 * M2_PHI_PITCH * 4 <= M2_TILE && M2_ATOMS * NF * 4 <= M2_TILE, "aliases: m over the operand tile, the fp32 tile over the hidden tile");

// tooling (-DMF_CLOCK_DEBUG): wall-clock stamps (100 MHz) of the first workgroup
struct M2Dbg {
#ifdef MF_CLOCK_DEBUG
  long long w[48];
  int n;
  __device__ __forceinline__ void init() { n = 0; w[n++] = wall_clock64(); }
  __device__ __forceinline__ void stamp() { if (n < 48) w[n++] = wall_clock64(); }
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

struct M2Frag {
  h8 hi[8], lo[8];
  float inv;
};
__device__ __forceinline__ void m2_pow2_scale(float m, float& S, float& inv) {
  const int e = (int)((__float_as_uint(m) >> 23) & 0xffu);
  const bool ok = e >= 40 && e < 255;
  S = ok ? __uint_as_float((unsigned)(268 - e) << 23) : 1.0f;
  inv = ok ? __uint_as_float((unsigned)(e - 14) << 23) : 1.0f;
}
// A fragments of output block nb from a fragment-order weight image (node128s.hip:load_wimg)
__device__ __forceinline__ void m2_load_w(M2Frag& w, const char* __restrict__ img, int nb, int r, int h) {
  const char* p = img + ((size_t)(nb * 8 * 64 + h * 32 + r) << 4);
#pragma unroll
  for (int T = 0; T < 8; ++T) {
    w.hi[T] = *reinterpret_cast<const h8*>(p + 1024 * T);
    w.lo[T] = *reinterpret_cast<const h8*>(p + M2_WIMG_PLANE + 1024 * T);
  }
  w.inv = *reinterpret_cast<const float*>(img + 2 * M2_WIMG_PLANE);
}
__device__ __forceinline__ float m2_amax16(const float (&v)[16]) {
  float m = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) m = fmaxf(m, fabsf(v[k]));
  return m;
}
// after the barrier that follows the publish: row scale, split, write this lane's 16 values of row r into the fragment-ordered planes
// v[4 q + c] = feature nb*32 + 8 q + 4 h + c  ->  k-slot (2 nb + (q >> 1)) * 16 + 8 h + 4 (q & 1) + c   (node128s.hip:tile_commit)
__device__ __forceinline__ float m2_commit(const float (&v)[16], char* tile, const float* pmax, int nb, int r, int h) {
  float m = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) m = fmaxf(m, pmax[j * 32 + r]);
  float S, inv;
  m2_pow2_scale(m, S, inv);
  char* row = tile + r * M2_PITCH + 2 * (nb * 32 + 8 * h);
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
    *reinterpret_cast<h4*>(row + M2_PLANE + off) = lo;
  }
  return inv;
}
// block nb of D^T = W . X^T for the 32 rows of the LDS tile (node128s.hip:tile_gemm_s)
__device__ __forceinline__ void m2_gemm(float (&out)[16], const char* tile, const M2Frag& w, float inv_row, int r, int h) {
  f32x16 acc;
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.f;
  const char* xr = tile + r * M2_PITCH + 16 * h;
  h8 bh0 = *reinterpret_cast<const h8*>(xr), bl0 = *reinterpret_cast<const h8*>(xr + M2_PLANE);
  h8 bh1 = *reinterpret_cast<const h8*>(xr + 32), bl1 = *reinterpret_cast<const h8*>(xr + M2_PLANE + 32);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int T = 0; T < 8; ++T) {
    h8 bh2, bl2;
    if (T < 6) {
      bh2 = *reinterpret_cast<const h8*>(xr + 32 * (T + 2));
      bl2 = *reinterpret_cast<const h8*>(xr + M2_PLANE + 32 * (T + 2));
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

// max over the 32 lanes of a half-wave, valid in lanes 31 / 63 (the DPP ladder of common.h:half_sum_top with fmaxf)
#define M2_DPP_MAX(v, ctrl, row_mask) \
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, row_mask, 0xF, false)))
__device__ __forceinline__ float m2_half_max_top(float v) {      // v >= 0
  M2_DPP_MAX(v, 0xB1, 0xF);
  M2_DPP_MAX(v, 0x4E, 0xF);
  M2_DPP_MAX(v, 0x141, 0xF);
  M2_DPP_MAX(v, 0x140, 0xF);
  M2_DPP_MAX(v, 0x142, 0xA);
  return v;
}
// m2_commit with the row maximum already known (written by the half-wave that produced the row): no publish / barrier round
__device__ __forceinline__ float m2_commit_known(const float (&v)[16], char* tile, float m, int nb, int r, int h) {
  float S, inv;
  m2_pow2_scale(m, S, inv);
  char* row = tile + r * M2_PITCH + 2 * (nb * 32 + 8 * h);
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
    *reinterpret_cast<h4*>(row + M2_PLANE + off) = lo;
  }
  return inv;
}

// -----------------------------------------------------------------------------------------------------------------------
// The order the persistent workgroups take molecules in: by the number of 32-pair tiles (what a molecule costs: the tile loops are
// 2/3 of a workgroup's time), LARGEST FIRST -- the longest-processing-time rule, so the launch ends within one small molecule of
// balanced.  A stable counting sort in one workgroup (molecules of equal tile count keep the caller's order: the schedule, and so the
// timing, is reproducible; the results never depend on it).  Also zeroes the head words of the step's launches.
// -----------------------------------------------------------------------------------------------------------------------
#define MO_THREADS 1024
#define MO_BINS 16
#define MO_QUEUES (2 * NNHIP_MAX_LAYERS)
__global__ void __launch_bounds__(MO_THREADS) mol2_order_kernel(const int* __restrict__ mol_ptr, const int* __restrict__ pair_ptr, int n_mol,
                                                                int* __restrict__ order, int* __restrict__ queue) {
  __shared__ int hist[MO_BINS], base[MO_BINS], wcnt[MO_THREADS / 64][MO_BINS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < MO_QUEUES) queue[tid] = 0;
  if (tid < MO_BINS) hist[tid] = 0;
  __syncthreads();
  auto key_of = [&](int b) {
    const int a0 = mol_ptr[b], a1 = mol_ptr[b + 1];
    const int np = a1 > a0 ? pair_ptr[a1] - pair_ptr[a0] : 0;
    const int t = (max(np, 0) + 31) >> 5;
    return MO_BINS - 1 - min(t, MO_BINS - 1);      // bin 0 = the most tiles
  };
  for (int b = tid; b < n_mol; b += MO_THREADS) atomicAdd(&hist[key_of(b)], 1);
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int k = 0; k < MO_BINS; ++k) {
      base[k] = run;
      run += hist[k];
    }
  }
  __syncthreads();
  for (int c0 = 0; c0 < n_mol; c0 += MO_THREADS) {
    const int b = c0 + tid;
    const int key = b < n_mol ? key_of(b) : -1;
    int rank = 0;
#pragma unroll
    for (int k = 0; k < MO_BINS; ++k) {
      const unsigned long long mask = __ballot(key == k);
      if (key == k) rank = __popcll(mask & ((1ull << lane) - 1ull));
      if (lane == 0) wcnt[wave][k] = __popcll(mask);
    }
    __syncthreads();
    if (key >= 0) {
      int off = base[key] + rank;
      for (int w = 0; w < wave; ++w) off += wcnt[w][key];
      order[off] = b;
    }
    __syncthreads();
    if (tid < MO_BINS) {
      int sum = 0;
      for (int w = 0; w < MO_THREADS / 64; ++w) sum += wcnt[w][tid];
      base[tid] += sum;
    }
    __syncthreads();
  }
}
// order: n_mol ints; queue: MO_QUEUES ints (launch k of the step uses queue + k)
int launch_mol2_order(const int* mol_ptr, const int* pair_ptr, int n_mol, int* order, int* queue, hipStream_t s) {
  ScopedTimer t0(TC_GRAPH, s);
  if (n_mol <= 0) return 0;
  mol2_order_kernel<<<1, MO_THREADS, 0, s>>>(mol_ptr, pair_ptr, n_mol, order, queue);
  LAUNCH_CHECK();
  return 0;
}

struct Mol2FwdArgs {
  const int *mol_ptr, *row_ptr, *pair_ptr, *col, *pid;
  const float* geo;
  const int2* xg;
  const float *m, *a_in, *f_in, *table;
  const char *img10, *img12, *img20, *img22;
  float *a_mid, *f_out, *h1, *h2, *phi1, *phi2, *msg;
  const int* order;   // molecules, largest first (mol2_order_kernel)
  int* queue;         // head word of this launch: molecules handed out beyond the first gridDim.x
  int n_mol;
};

template <bool HAS_F>
__global__ void __launch_bounds__(M2_THREADS, 2) mol2_edge_fwd_kernel(const Mol2FwdArgs A) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* xt = lds;
  char* ht = lds + M2_OFF_H;
  float* phit = reinterpret_cast<float*>(lds + M2_OFF_H);
  float* pmax = reinterpret_cast<float*>(lds + M2_OFF_PMAX);
  float* sm_m = reinterpret_cast<float*>(lds);                       // pass 1 only
  float* sm_f = reinterpret_cast<float*>(lds + M2_OFF_F);
  float4* sm_geo = reinterpret_cast<float4*>(lds + M2_OFF_GEO);
  int2* sm_xg = reinterpret_cast<int2*>(lds + M2_OFF_XG);
  int* sm_rowb = reinterpret_cast<int*>(lds + M2_OFF_ROWB);
  unsigned short* sm_pij = reinterpret_cast<unsigned short*>(lds + M2_OFF_PIJ);   // [pair] i | j << 8 (molecule-local)

  int* sm_q = reinterpret_cast<int*>(lds + M2_OFF_Q);
  const int tid0 = threadIdx.x;

  M2Dbg dbg;
  dbg.init();
  M2Frag w1, w2;
  if (!HAS_F) {             // layer 0 runs one edge MLP: its fragments stay in the registers for every molecule of the launch
    const int lane0 = tid0 & 63;
    m2_load_w(w1, A.img10, tid0 >> 6, lane0 & 31, lane0 >> 5);
    m2_load_w(w2, A.img12, tid0 >> 6, lane0 & 31, lane0 >> 5);
  }
  // ---- one molecule (everything the kernel did per workgroup before round 6)
  auto molecule = [&](const int b) {
  // (the thread index passes through an empty asm once per molecule: hoisted out of the queue loop, the per-lane address arithmetic of
  // the whole body would live across it -- 165 spilled registers instead of 20)
  int tid = tid0;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63, nb = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5, c4 = 4 * r, c2 = 2 * lane;
  const int a0 = A.mol_ptr[b], n = A.mol_ptr[b + 1] - a0;
  if (n <= 0 || n > M2_ATOMS) return;                                // (uniform: a workgroup serves whole molecules)
  const int E0 = A.row_ptr[a0], nE = A.row_ptr[a0 + n] - E0;
  const int P0 = A.pair_ptr[a0], nP = A.pair_ptr[a0 + n] - P0;
  if (nE < 0 || nE > M2_EDGES || nP < 0 || nP > M2_PAIRS || nE != 2 * nP) return;
  const int nT = (nP + 31) >> 5;
  // ---- the first MLP's fragments are requested before anything else (they arrive under the list and message passes)
  if (HAS_F) {
    m2_load_w(w1, A.img10, nb, r, h);
    m2_load_w(w2, A.img12, nb, r, h);
  }
  __builtin_amdgcn_sched_barrier(0);

  // ---- lists: wave w walks the rows of ITS atoms (w, w + 4, ...), a lane per edge.  The row's incidence descriptors stay in
  // the wave's registers (inc_k[k], lane l = edge l of the row; rt_k[k], lane t = how many of the row's pairs lie below tile t):
  // the sums below fetch them with v_readlane -- no LDS round trip between an incidence and its rows.
  if (tid <= n) sm_rowb[tid] = A.row_ptr[a0 + tid] - E0;
  {
    const float4* s = reinterpret_cast<const float4*>(A.m + (size_t)a0 * NF);
    for (int t = tid; t < n * 32; t += M2_THREADS) reinterpret_cast<float4*>(sm_m)[t] = s[t];
    if (HAS_F) {
      const float4* sf = reinterpret_cast<const float4*>(A.f_in + (size_t)a0 * 3 * NF);
      for (int t = tid; t < n * 96; t += M2_THREADS) reinterpret_cast<float4*>(sm_f)[t] = sf[t];
    }
  }
  int inc_k[M2_OWN], rt_k[M2_OWN];
  {
    // (three rounds of independent loads: extents of the wave's six rows; their col / pid / geo / xg entries; then the arithmetic)
    int beg_k[M2_OWN], deg_k[M2_OWN], col_k[M2_OWN], pid_k[M2_OWN];
    float4 geo_k[M2_OWN];
    int2 xg_k[M2_OWN];
#pragma unroll
    for (int k = 0; k < M2_OWN; ++k) {
      const int a = nb + M2_WAVES * k;
      beg_k[k] = a < n ? A.row_ptr[a0 + a] : 0;
      deg_k[k] = a < n ? A.row_ptr[a0 + a + 1] - beg_k[k] : 0;
    }
#pragma unroll
    for (int k = 0; k < M2_OWN; ++k) {
      const int e = beg_k[k] + (lane < deg_k[k] ? lane : 0);
      const bool on = lane < deg_k[k];
      col_k[k] = on ? A.col[e] : 0;
      pid_k[k] = on ? A.pid[e] : 0;
      geo_k[k] = on ? reinterpret_cast<const float4*>(A.geo)[e] : make_float4(0.f, 0.f, 0.f, 1.f);
      xg_k[k] = on ? A.xg[e] : make_int2(0, 0);
    }
#pragma unroll
    for (int k = 0; k < M2_OWN; ++k) {
      const int a = nb + M2_WAVES * k;
      int inc = 0x7fff, rt = 0, p = 0x3fff;
      if (lane < deg_k[k]) {
        int j = col_k[k] - a0;
        p = pid_k[k] - P0;
        j = min(max(j, 0), n - 1);              // (a valid list never needs these; they keep every LDS index inside its array)
        p = min(max(p, 0), max(nP - 1, 0));
        const bool own = j > a;
        inc = p | (j << 9) | (own ? (1 << 14) : 0);
        if (own) {
          sm_geo[p] = geo_k[k];
          sm_xg[p] = xg_k[k];
          sm_pij[p] = (unsigned short)(a | (j << 8));
        }
      }
      if (a < n) {
#pragma unroll
        for (int t = 0; t <= M2_MAX_TILES; ++t) {
          const int c = __popcll(__ballot(p < 32 * t));
          if (lane == t) rt = c;
        }
      }
      inc_k[k] = inc;
      rt_k[k] = rt;
    }
  }
  __syncthreads();
  dbg.stamp();
  // ---- pass 1: the messages, once per pair (a half-wave per pair, four pairs in flight: msg = eps * m[i] * m[j], newtonnet.py:210-211),
  // a tile of 32 pairs at a time: rows to the L2 (the MLPs read them back tile by tile) and into an fp32 tile behind m, from which
  // a_mid = a_in + sum of the messages of a's edges (newtonnet.py:213-215) is formed by the owner waves.  The radial-filter rows
  // of the NEXT tile are requested before this tile's sums (they fly under them).
  {
    float2 acc_a[M2_OWN];
#pragma unroll
    for (int k = 0; k < M2_OWN; ++k) acc_a[k] = make_float2(0.f, 0.f);
    float4 trow[4][4];
    FilterW fw[4];
    auto request = [&](int t) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pl = 32 * t + 8 * u + 2 * nb + h;
        const int2 gx = sm_xg[pl < nP ? pl : 0];
        fw[u] = filter_weights(__int_as_float(gx.y));
#pragma unroll
        for (int q = 0; q < 4; ++q) trow[u][q] = ld4(A.table + (size_t)(gx.x + q) * NF + c4);    // (edge_common.h:filter_value)
      }
    };
    if (nT > 0) request(0);
#pragma unroll 1
    for (int t = 0; t < nT; ++t) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pl = 32 * t + 8 * u + 2 * nb + h;
        if (pl < nP) {
          float4 eps = mul4(trow[u][0], fw[u].w[0]);
#pragma unroll
          for (int q = 1; q < 4; ++q) eps = fma4(trow[u][q], fw[u].w[q], eps);
          const int ij = sm_pij[pl];
          const float4 v = mul4(mul4(eps, *reinterpret_cast<const float4*>(sm_m + (ij & 255) * NF + c4)),
                                *reinterpret_cast<const float4*>(sm_m + (ij >> 8) * NF + c4));
          st4(A.msg + (size_t)(P0 + pl) * NF + c4, v);
          *reinterpret_cast<float4*>(phit + (pl - 32 * t) * M2_PHI_PITCH + c4) = v;
          // the row's largest magnitude, for the MLPs' operand scale (geo.w -- r -- is not used by this kernel)
          const float mx = m2_half_max_top(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
          if (r == 31) reinterpret_cast<float*>(sm_geo + pl)[3] = mx;
        }
      }
      __syncthreads();
      if (t + 1 < nT) request(t + 1);
#pragma unroll
      for (int k = 0; k < M2_OWN; ++k) {
        const int a = nb + M2_WAVES * k;
        if (a < n) {
          const int eb = __builtin_amdgcn_readlane(rt_k[k], t), ee = __builtin_amdgcn_readlane(rt_k[k], t + 1);
          for (int l = eb; l < ee; l += M2_AGG) {      // (M2_AGG incidences in flight: one LDS round trip serves most atoms)
            float2 v[M2_AGG];
#pragma unroll
            for (int u = 0; u < M2_AGG; ++u) {
              const bool on = l + u < ee;
              const int iu = __builtin_amdgcn_readlane(inc_k[k], on ? l + u : l);
              v[u] = *reinterpret_cast<const float2*>(phit + ((iu & 511) - 32 * t) * M2_PHI_PITCH + c2);
              if (!on) v[u] = make_float2(0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < M2_AGG; ++u) acc_a[k] = acc_a[k] + v[u];
          }
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < M2_OWN; ++k) {
      const int a = nb + M2_WAVES * k;
      if (a < n) st2(A.a_mid + (size_t)(a0 + a) * NF + c2, ld2(A.a_in + (size_t)(a0 + a) * NF + c2) + acc_a[k]);
    }
  }
  __syncthreads();          // msg rows written (this workgroup reads them back below), m is dead: the tiles take its place
  dbg.stamp();

  // ---- pass 2: the edge MLPs, tile by tile, and the force-message sums
  auto run_mlp = [&](auto MLP_) {
    constexpr int mlp = decltype(MLP_)::value;     // (compile-time: each instantiation carries only its own sums)
    if (mlp == 1) {
      m2_load_w(w1, A.img20, nb, r, h);
      m2_load_w(w2, A.img22, nb, r, h);
    }
    float* Hk = mlp ? A.h2 : A.h1;
    float* Y = mlp ? A.phi2 : A.phi1;
    float2 acc[M2_OWN][3];
#pragma unroll
    for (int k = 0; k < M2_OWN; ++k)
#pragma unroll
      for (int q = 0; q < 3; ++q) acc[k][q] = make_float2(0.f, 0.f);
    // this lane's 16 values of a row: features nb*32 + 8 q + 4 h + c
    float x[16];
    auto load_x = [&](int t) {
      const int pl = 32 * t + r;
      const float4* xp = reinterpret_cast<const float4*>(A.msg + (size_t)(P0 + min(pl, max(nP - 1, 0))) * NF + nb * 32 + 4 * h);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = (pl < nP) ? xp[2 * q] : make_float4(0.f, 0.f, 0.f, 0.f);
        x[4 * q] = v.x, x[4 * q + 1] = v.y, x[4 * q + 2] = v.z, x[4 * q + 3] = v.w;
      }
    };
    if (nT > 0) load_x(0);
#pragma unroll 1
    for (int t = 0; t < nT; ++t) {
      const int pl = 32 * t + r;
      const bool live = pl < nP;
      const size_t pg = (size_t)P0 + pl;
      const size_t tile_g = pg >> 5;
      const int lane_g = 32 * h + (int)(pg & 31);
      // operand tile of stage 1
      if (t < 2) dbg.stamp();
      // (every wave is past stage 1 of the previous tile -- four barriers ago -- so the operand tile is free; the row maxima were
      // left in geo.w by the message pass: no publish / barrier round for this commit)
#if M2_KNOWN_MAX
      const float inv_x = m2_commit_known(x, xt, live ? sm_geo[32 * t + r].w : 0.f, nb, r, h);
#else
      pmax[(nb * 2 + h) * 32 + r] = m2_amax16(x);
      __syncthreads();
      const float inv_x = m2_commit(x, xt, pmax, nb, r, h);
#endif
      __syncthreads();
      if (t < 2) dbg.stamp();
      float hv[16];
      m2_gemm(hv, xt, w1, inv_x, r, h);
      // keep silu'(h) for the adjoint (fragment order of mlp128s.hip, by global pair tile), activate
      {
        float4* hp = reinterpret_cast<float4*>(Hk) + (tile_g * 4 + nb) * 256 + lane_g;
        float keep[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const float v = hv[k], s = sigmoid_f(v);
          keep[k] = s * (1.0f + v * (1.0f - s));
          hv[k] = v * s;
        }
        if (live) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            st4_nt(reinterpret_cast<float*>(hp + 64 * q), make_float4(keep[4 * q], keep[4 * q + 1], keep[4 * q + 2], keep[4 * q + 3]));
        }
      }
      if (t < 2) dbg.stamp();
      pmax[(nb * 2 + h) * 32 + r] = m2_amax16(hv);
      __syncthreads();                                    // (also: every wave is past its last read of the previous phi tile)
      const float inv_h = m2_commit(hv, ht, pmax, nb, r, h);
      __syncthreads();
      if (t < 2) dbg.stamp();
      float y[16];
      m2_gemm(y, ht, w2, inv_h, r, h);
      if (live) {
        float4* yp = reinterpret_cast<float4*>(Y + pg * NF + nb * 32 + 4 * h);
#pragma unroll
        for (int q = 0; q < 4; ++q) yp[2 * q] = make_float4(y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]);
      }
      __syncthreads();                                    // every wave is done reading the hidden tile: phi takes its place
      {
        float* yr = phit + r * M2_PHI_PITCH + nb * 32 + 4 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<float4*>(yr + 8 * q) = make_float4(y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]);
      }
      __syncthreads();
      if (t < 2) dbg.stamp();
      if (t + 1 < nT) load_x(t + 1);                     // (the next tile's rows fly under this tile's sums; x is dead since its commit)
      // force-message sums of this tile: a full wave per incidence (two features per lane), the wave's atoms one after the other,
      // two incidences in flight; the descriptors come out of the wave's own registers (v_readlane)
#pragma unroll
      for (int k = 0; k < M2_OWN; ++k) {
        const int a = nb + M2_WAVES * k;
        if (a < n) {
          const int eb = __builtin_amdgcn_readlane(rt_k[k], t), ee = __builtin_amdgcn_readlane(rt_k[k], t + 1);
          for (int l = eb; l < ee; l += M2_AGG) {
            float2 v[M2_AGG];
            int iu[M2_AGG];
#pragma unroll
            for (int u = 0; u < M2_AGG; ++u) {
              const bool on = l + u < ee;
              iu[u] = __builtin_amdgcn_readlane(inc_k[k], on ? l + u : l);
              v[u] = *reinterpret_cast<const float2*>(phit + ((iu[u] & 511) - 32 * t) * M2_PHI_PITCH + c2);
              if (!on) v[u] = make_float2(0.f, 0.f);
            }
            if (mlp == 0) {
              float4 g[M2_AGG];
#pragma unroll
              for (int u = 0; u < M2_AGG; ++u) g[u] = sm_geo[iu[u] & 511];
#pragma unroll
              for (int u = 0; u < M2_AGG; ++u) {
                const float sg = (iu[u] >> 14) & 1 ? 1.0f : -1.0f;       // u of the reverse direction is -u
                acc[k][0] = fma2(v[u], sg * g[u].x, acc[k][0]);
                acc[k][1] = fma2(v[u], sg * g[u].y, acc[k][1]);
                acc[k][2] = fma2(v[u], sg * g[u].z, acc[k][2]);
              }
            } else {
              float2 fj[M2_AGG][3];
#pragma unroll
              for (int u = 0; u < M2_AGG; ++u)
#pragma unroll
                for (int q = 0; q < 3; ++q) fj[u][q] = *reinterpret_cast<const float2*>(sm_f + (((iu[u] >> 9) & 31) * 3 + q) * NF + c2);
#pragma unroll
              for (int u = 0; u < M2_AGG; ++u)
#pragma unroll
                for (int q = 0; q < 3; ++q) acc[k][q] = fma2(v[u], fj[u][q], acc[k][q]);
            }
          }
        }
      }
    }
    dbg.stamp();
    // f_out = f_in + sum (first MLP), += sum (second MLP: what this very lane wrote)
#pragma unroll
    for (int k = 0; k < M2_OWN; ++k) {
      const int a = nb + M2_WAVES * k;
      if (a < n) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          float* fo = A.f_out + ((size_t)(a0 + a) * 3 + q) * NF + c2;
          float2 base = make_float2(0.f, 0.f);
          if (mlp == 1)
            base = ld2(fo);
          else if (HAS_F)
            base = *reinterpret_cast<const float2*>(sm_f + (a * 3 + q) * NF + c2);
          st2(fo, base + acc[k][q]);
        }
      }
    }
  };
  run_mlp(std::integral_constant<int, 0>());
  if (HAS_F) run_mlp(std::integral_constant<int, 1>());
  dbg.stamp();
  };
  // ---- the queue: position q of the order; the first gridDim.x positions are the workgroups' own, the head word hands out the
  // rest.  The atomic for the NEXT molecule is issued before this one starts and its value is looked at when this one is done.
  int it = 0;
  for (int q = blockIdx.x; q < A.n_mol; ++it) {
    int nxt = 0;
    if (tid0 == 0) nxt = (int)gridDim.x + atomicAdd(A.queue, 1);
    molecule(A.order[q]);
    if (tid0 == 0) sm_q[it & 1] = nxt;
    __syncthreads();          // (also: every wave is done with this molecule's LDS before the next one's lists land)
    q = sm_q[it & 1];
#ifdef MF_CLOCK_DEBUG
    if (it == 0) dbg.print(HAS_F ? "mol2_fwd<1>" : "mol2_fwd<0>");
#endif
  }
}

int launch_mol2_edge_fwd(bool has_f, const int* mol_ptr, const int* row_ptr, const int* pair_ptr, const int* col, const int* pid,
                         const float* geo, const int* xg, const float* m, const float* a_in, const float* f_in, const float* table,
                         const char* img10, const char* img12, const char* img20, const char* img22, float* a_mid, float* f_out,
                         float* h1, float* h2, float* phi1, float* phi2, float* msg, const int* order, int* queue, int n_mol,
                         hipStream_t s) {
  ScopedTimer t0(TC_MOL_FWD, s);
  static const hipError_t rc0 = hipFuncSetAttribute((const void*)mol2_edge_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, M2_LDS);
  static const hipError_t rc1 = hipFuncSetAttribute((const void*)mol2_edge_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, M2_LDS);
  HIP_TRY(rc0);
  HIP_TRY(rc1);
  if (n_mol <= 0) return 0;
  Mol2FwdArgs A = {mol_ptr, row_ptr, pair_ptr, col, pid, geo, reinterpret_cast<const int2*>(xg), m, a_in, f_in, table,
                   img10, img12, img20, img22, a_mid, f_out, h1, h2, phi1, phi2, msg, order, queue, n_mol};
  const int grid = n_mol < M2_GRID ? n_mol : M2_GRID;
#ifdef M2_DBG_ONE_PER_CU   // tooling: 100 KB of LDS per workgroup = one workgroup per CU
  const size_t lds_bytes = 102400;
  (void)hipFuncSetAttribute((const void*)mol2_edge_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  (void)hipFuncSetAttribute((const void*)mol2_edge_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
#else
  const size_t lds_bytes = M2_LDS;
#endif
  if (has_f)
    mol2_edge_fwd_kernel<true><<<grid, M2_THREADS, lds_bytes, s>>>(A);
  else
    mol2_edge_fwd_kernel<false><<<grid, M2_THREADS, lds_bytes, s>>>(A);
  LAUNCH_CHECK();
  return 0;
}

// -----------------------------------------------------------------------------------------------------------------------
// adjoint, same organisation (4-wave workgroups, two per CU).  Passes of one workgroup, each inside 80 KB of LDS:
//   B1a  gf staged; per pair tile the kept phi1 / phi2 rows staged: g_u of both directions, g_phi1 rows (to the L2) and
//        g_fin = gf + sum phi2 * gf[j] (owner waves, registers)                                  (edge.hip:force_bwd_kernel)
//   B1b  f_in staged next to gf: g_phi2 rows (to the L2)
//   B2   per MLP, tile by tile: (g_phi_k W_k2) * silu'(h_k) W_k0 with the weight fragments in registers; the first MLP's term goes
//        to the L2 and is added by the second one's last stage                                   (mlp128s.hip, adjoint mode)
//   B3   inside the last MLP's tile loop: G = g_msg + g_a[i] + g_a[j] through an fp32 tile -> g_x (Hermite derivative of the radial
//        filter), G eps -> g_m sums (owner waves, registers)                                     (edge.hip:msg_bwd_kernel)
// -----------------------------------------------------------------------------------------------------------------------
#define M2B_OFF_F (3 * M2_ATOMS * NF * 4)                  // f_in behind gf (B1b); the two phi tiles of B1a lie here too
#define M2B_OFF_PHI2 (M2B_OFF_F + 32 * M2_PHI_PITCH * 4)
#define M2B_OFF_LIST (2 * 3 * M2_ATOMS * NF * 4)           // 73 728: geo | xg | pij | rowb | pairb
#define M2B_OFF_XG (M2B_OFF_LIST + M2_PAIRS * 16)
#define M2B_OFF_PIJ (M2B_OFF_XG + M2_PAIRS * 8)
#define M2B_OFF_ROWB (M2B_OFF_PIJ + M2_PAIRS * 2)
#define M2B_OFF_PAIRB (M2B_OFF_ROWB + (M2_ATOMS + 1) * 4)
#define M2B_OFF_Q (M2B_OFF_PAIRB + (M2_ATOMS + 1) * 4 + 8)   // two queue slots, as the forward
#define M2B_LDS (M2B_OFF_Q + 8)
#define M2B_OFF_M (2 * M2_TILE + 8 * 32 * 4)               // B2: operand tile | hidden / fp32 tile | row maxima | m | g_a
#define M2B_OFF_GA (M2B_OFF_M + M2_ATOMS * NF * 4)
static_assert(M2B_LDS <= 81920, "two workgroups per CU");
static_assert(M2B_OFF_PHI2 + 32 * M2_PHI_PITCH * 4 <= M2B_OFF_LIST && M2B_OFF_GA + M2_ATOMS * NF * 4 <= M2B_OFF_LIST, "overlays");

struct Mol2BwdArgs {
  const int *mol_ptr, *row_ptr, *pair_ptr, *col, *pid, *rev;
  const float* geo;
  const int2* xg;
  const float *gf, *g_a, *m, *f_in, *table;
  const char *img12T, *img10T, *img22T, *img20T;
  const float *h1, *h2, *phi1, *phi2;
  float *g_fin, *g_m, *g_x, *g_u;
  float* g_phi;     // [P][256] scratch: g_phi1 | g_phi2 rows between B1 and B2 (the row path's g_h12 array)
  float* g_msg;     // [P][128] scratch: the first MLP's term of g_msg
  const int* order;
  int* queue;
  int n_mol;
};

template <bool LOWER>
__global__ void __launch_bounds__(M2_THREADS, 2) mol2_edge_bwd_kernel(const Mol2BwdArgs A) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float* sm_gf = reinterpret_cast<float*>(lds);
  float* sm_f = reinterpret_cast<float*>(lds + M2B_OFF_F);
  float* phi1t = reinterpret_cast<float*>(lds + M2B_OFF_F);
  float* phi2t = reinterpret_cast<float*>(lds + M2B_OFF_PHI2);
  float4* sm_geo = reinterpret_cast<float4*>(lds + M2B_OFF_LIST);
  int2* sm_xg = reinterpret_cast<int2*>(lds + M2B_OFF_XG);
  unsigned short* sm_pij = reinterpret_cast<unsigned short*>(lds + M2B_OFF_PIJ);
  int* sm_rowb = reinterpret_cast<int*>(lds + M2B_OFF_ROWB);
  int* sm_pairb = reinterpret_cast<int*>(lds + M2B_OFF_PAIRB);
  char* xt = lds;
  char* ht = lds + M2_OFF_H;
  float* phit = reinterpret_cast<float*>(lds + M2_OFF_H);
  float* pmax = reinterpret_cast<float*>(lds + M2_OFF_PMAX);
  float* sm_m = reinterpret_cast<float*>(lds + M2B_OFF_M);
  float* sm_ga = reinterpret_cast<float*>(lds + M2B_OFF_GA);

  int* sm_q = reinterpret_cast<int*>(lds + M2B_OFF_Q);
  const int tid0 = threadIdx.x;
  M2Dbg dbg;
  dbg.init();
  M2Frag w1, w2;
  if (!LOWER) {             // layer 0 has one edge MLP: its adjoint's fragments are loaded once per launch
    const int lane0 = tid0 & 63;
    m2_load_w(w1, A.img12T, tid0 >> 6, lane0 & 31, lane0 >> 5);
    m2_load_w(w2, A.img10T, tid0 >> 6, lane0 & 31, lane0 >> 5);
  }
  auto molecule = [&](const int b) {
  int tid = tid0;             // (laundered once per molecule, as the forward)
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63, nb = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5, c4 = 4 * r, c2 = 2 * lane;
  const int a0 = A.mol_ptr[b], n = A.mol_ptr[b + 1] - a0;
  if (n <= 0 || n > M2_ATOMS) return;
  const int E0 = A.row_ptr[a0], nE = A.row_ptr[a0 + n] - E0;
  const int P0 = A.pair_ptr[a0], nP = A.pair_ptr[a0 + n] - P0;
  if (nE < 0 || nE > M2_EDGES || nP < 0 || nP > M2_PAIRS || nE != 2 * nP) return;
  const int nT = (nP + 31) >> 5;

  // ---- lists (as the forward) + gf
  if (tid <= n) {
    sm_rowb[tid] = A.row_ptr[a0 + tid] - E0;
    sm_pairb[tid] = A.pair_ptr[a0 + tid] - P0;
  }
  {
    const float4* s = reinterpret_cast<const float4*>(A.gf + (size_t)a0 * 3 * NF);
    for (int t = tid; t < n * 96; t += M2_THREADS) reinterpret_cast<float4*>(sm_gf)[t] = s[t];
  }
  int inc_k[M2_OWN], rt_k[M2_OWN];
  {
    int beg_k[M2_OWN], deg_k[M2_OWN], col_k[M2_OWN], pid_k[M2_OWN];
    float4 geo_k[M2_OWN];
    int2 xg_k[M2_OWN];
#pragma unroll
    for (int k = 0; k < M2_OWN; ++k) {
      const int a = nb + M2_WAVES * k;
      beg_k[k] = a < n ? A.row_ptr[a0 + a] : 0;
      deg_k[k] = a < n ? A.row_ptr[a0 + a + 1] - beg_k[k] : 0;
    }
#pragma unroll
    for (int k = 0; k < M2_OWN; ++k) {
      const int e = beg_k[k] + (lane < deg_k[k] ? lane : 0);
      const bool on = lane < deg_k[k];
      col_k[k] = on ? A.col[e] : 0;
      pid_k[k] = on ? A.pid[e] : 0;
      geo_k[k] = on ? reinterpret_cast<const float4*>(A.geo)[e] : make_float4(0.f, 0.f, 0.f, 1.f);
      xg_k[k] = on ? A.xg[e] : make_int2(0, 0);
    }
#pragma unroll
    for (int k = 0; k < M2_OWN; ++k) {
      const int a = nb + M2_WAVES * k;
      int inc = 0x7fff, rt = 0, p = 0x3fff;
      if (lane < deg_k[k]) {
        int j = col_k[k] - a0;
        p = pid_k[k] - P0;
        j = min(max(j, 0), n - 1);
        p = min(max(p, 0), max(nP - 1, 0));
        const bool own = j > a;
        inc = p | (j << 9) | (own ? (1 << 14) : 0);
        if (own) {
          sm_geo[p] = geo_k[k];
          sm_xg[p] = xg_k[k];
          sm_pij[p] = (unsigned short)(a | (j << 8));
        }
      }
      if (a < n) {
#pragma unroll
        for (int t = 0; t <= M2_MAX_TILES; ++t) {
          const int c = __popcll(__ballot(p < 32 * t));
          if (lane == t) rt = c;
        }
      }
      inc_k[k] = inc;
      rt_k[k] = rt;
    }
  }
  // the owner's directed edge of pair p (rows list their own pairs last, in pair order)
  auto owner_edge = [&](int p, int i) { return E0 + sm_rowb[i + 1] - (sm_pairb[i + 1] - p); };
  __syncthreads();
  dbg.stamp();

  // ---- B1a: per tile, the kept phi rows staged; g_u, g_phi1, g_fin
  {
    float2 acc[M2_OWN][3];
#pragma unroll
    for (int k = 0; k < M2_OWN; ++k)
#pragma unroll
      for (int q = 0; q < 3; ++q) acc[k][q] = make_float2(0.f, 0.f);
    float4 s1[4], s2[4];
    auto request = [&](int t) {          // 32 rows x 32 float4 per array, four per thread
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = tid + M2_THREADS * u, row = idx >> 5;
        const bool on = 32 * t + row < nP;
        const size_t g = ((size_t)P0 + 32 * t + (on ? row : 0)) * (NF / 4) + (idx & 31);
        s1[u] = on ? reinterpret_cast<const float4*>(A.phi1)[g] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (LOWER) s2[u] = on ? reinterpret_cast<const float4*>(A.phi2)[g] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    };
    if (nT > 0) request(0);
#pragma unroll 1
    for (int t = 0; t < nT; ++t) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = tid + M2_THREADS * u;
        *reinterpret_cast<float4*>(phi1t + (idx >> 5) * M2_PHI_PITCH + 4 * (idx & 31)) = s1[u];
        if (LOWER) *reinterpret_cast<float4*>(phi2t + (idx >> 5) * M2_PHI_PITCH + 4 * (idx & 31)) = s2[u];
      }
      __syncthreads();
      if (t + 1 < nT) request(t + 1);
      // g_u of both directions and the g_phi1 row of every pair of the tile (a half-wave per pair)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pl = 32 * t + 8 * u + 2 * nb + h;
        if (pl < nP) {
          const int ij = sm_pij[pl], i = ij & 255, j = ij >> 8;
          const float4 g = sm_geo[pl];
          const float4 v1 = *reinterpret_cast<const float4*>(phi1t + (pl - 32 * t) * M2_PHI_PITCH + c4);
          float si[3], sj[3];
          float4 gp = make_float4(0.f, 0.f, 0.f, 0.f);
          const float uk[3] = {g.x, g.y, g.z};
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const float4 gi = *reinterpret_cast<const float4*>(sm_gf + (i * 3 + k) * NF + c4);
            const float4 gj = *reinterpret_cast<const float4*>(sm_gf + (j * 3 + k) * NF + c4);
            si[k] = half_sum_top(dot4(gi, v1));
            sj[k] = half_sum_top(dot4(gj, v1));
            gp = fma4(sub4(gi, gj), uk[k], gp);
          }
          st4(A.g_phi + ((size_t)P0 + pl) * 2 * NF + c4, gp);
          const float mx = m2_half_max_top(fmaxf(fmaxf(fabsf(gp.x), fabsf(gp.y)), fmaxf(fabsf(gp.z), fabsf(gp.w))));
          if (r == 31) {
            reinterpret_cast<float*>(sm_geo + pl)[3] = mx;      // the row's largest magnitude: the first MLP's operand scale (geo.w is free)
            const int e = owner_edge(pl, i);
            reinterpret_cast<float4*>(A.g_u)[e] = make_float4(si[0], si[1], si[2], 0.f);
            reinterpret_cast<float4*>(A.g_u)[A.rev[e]] = make_float4(sj[0], sj[1], sj[2], 0.f);
          }
        }
      }
      if (LOWER) {
#pragma unroll
        for (int k = 0; k < M2_OWN; ++k) {
          const int a = nb + M2_WAVES * k;
          if (a < n) {
            const int eb = __builtin_amdgcn_readlane(rt_k[k], t), ee = __builtin_amdgcn_readlane(rt_k[k], t + 1);
            for (int l = eb; l < ee; l += M2_AGG) {
              float2 v[M2_AGG], gj[M2_AGG][3];
#pragma unroll
              for (int u = 0; u < M2_AGG; ++u) {
                const bool on = l + u < ee;
                const int iu = __builtin_amdgcn_readlane(inc_k[k], on ? l + u : l);
                v[u] = *reinterpret_cast<const float2*>(phi2t + ((iu & 511) - 32 * t) * M2_PHI_PITCH + c2);
                if (!on) v[u] = make_float2(0.f, 0.f);
#pragma unroll
                for (int q = 0; q < 3; ++q) gj[u][q] = *reinterpret_cast<const float2*>(sm_gf + (((iu >> 9) & 31) * 3 + q) * NF + c2);
              }
#pragma unroll
              for (int u = 0; u < M2_AGG; ++u)
#pragma unroll
                for (int q = 0; q < 3; ++q) acc[k][q] = fma2(v[u], gj[u][q], acc[k][q]);
            }
          }
        }
      }
      __syncthreads();
    }
    if (LOWER) {
#pragma unroll
      for (int k = 0; k < M2_OWN; ++k) {
        const int a = nb + M2_WAVES * k;
        if (a < n) {
#pragma unroll
          for (int q = 0; q < 3; ++q)
            st2(A.g_fin + ((size_t)(a0 + a) * 3 + q) * NF + c2, *reinterpret_cast<const float2*>(sm_gf + (a * 3 + q) * NF + c2) + acc[k][q]);
        }
      }
    }
  }
  dbg.stamp();
  // ---- B1b: g_phi2[p] = sum_k gf[i][k] * f_in[j][k] + gf[j][k] * f_in[i][k]
  if (LOWER) {
    const float4* s = reinterpret_cast<const float4*>(A.f_in + (size_t)a0 * 3 * NF);
    for (int t = tid; t < n * 96; t += M2_THREADS) reinterpret_cast<float4*>(sm_f)[t] = s[t];
    __syncthreads();
    for (int pl = 2 * nb + h; pl < nP; pl += 2 * M2_WAVES) {
      const int ij = sm_pij[pl], i = ij & 255, j = ij >> 8;
      float4 gp = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        gp = fma4(*reinterpret_cast<const float4*>(sm_gf + (i * 3 + k) * NF + c4), *reinterpret_cast<const float4*>(sm_f + (j * 3 + k) * NF + c4), gp);
        gp = fma4(*reinterpret_cast<const float4*>(sm_gf + (j * 3 + k) * NF + c4), *reinterpret_cast<const float4*>(sm_f + (i * 3 + k) * NF + c4), gp);
      }
      st4(A.g_phi + ((size_t)P0 + pl) * 2 * NF + NF + c4, gp);
    }
  }
  __syncthreads();          // the g_phi rows are written (read back below), gf / f_in are dead
  dbg.stamp();

  // ---- B2 / B3
  {
    const float4* sm_ = reinterpret_cast<const float4*>(A.m + (size_t)a0 * NF);
    const float4* sg_ = reinterpret_cast<const float4*>(A.g_a + (size_t)a0 * NF);
    for (int t = tid; t < n * 32; t += M2_THREADS) {
      reinterpret_cast<float4*>(sm_m)[t] = sm_[t];
      reinterpret_cast<float4*>(sm_ga)[t] = sg_[t];
    }
  }
  float2 acc_m[M2_OWN];
#pragma unroll
  for (int k = 0; k < M2_OWN; ++k) acc_m[k] = make_float2(0.f, 0.f);
  auto run_mlp = [&](auto MLP_, auto LAST_) {
    constexpr int mlp = decltype(MLP_)::value;
    constexpr bool last = decltype(LAST_)::value;      // the MLP whose second stage completes g_msg: the message adjoint follows per tile
    if (LOWER) {
      m2_load_w(w1, mlp ? A.img22T : A.img12T, nb, r, h);
      m2_load_w(w2, mlp ? A.img20T : A.img10T, nb, r, h);
    }
    const float* Hk = mlp ? A.h2 : A.h1;
    float x[16];
    auto load_x = [&](int t) {
      const int pl = 32 * t + r;
      const float4* xp = reinterpret_cast<const float4*>(A.g_phi + ((size_t)P0 + min(pl, max(nP - 1, 0))) * 2 * NF + mlp * NF + nb * 32 + 4 * h);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = (pl < nP) ? xp[2 * q] : make_float4(0.f, 0.f, 0.f, 0.f);
        x[4 * q] = v.x, x[4 * q + 1] = v.y, x[4 * q + 2] = v.z, x[4 * q + 3] = v.w;
      }
    };
    if (nT > 0) load_x(0);
#pragma unroll 1
    for (int t = 0; t < nT; ++t) {
      const int pl = 32 * t + r;
      const bool live = pl < nP;
      const size_t pg = (size_t)P0 + pl;
      const size_t tile_g = pg >> 5;
      const int lane_g = 32 * h + (int)(pg & 31);
      float inv_x;
      if (mlp == 0) {       // (row maxima of g_phi1 left in geo.w by B1a: no exchange)
        inv_x = m2_commit_known(x, xt, live ? sm_geo[32 * t + r].w : 0.f, nb, r, h);
      } else {
        pmax[(nb * 2 + h) * 32 + r] = m2_amax16(x);
        __syncthreads();
        inv_x = m2_commit(x, xt, pmax, nb, r, h);
      }
      __syncthreads();
      float4 hin[4];
      {
        const float4* hp = reinterpret_cast<const float4*>(Hk) + (tile_g * 4 + nb) * 256 + lane_g;
#pragma unroll
        for (int q = 0; q < 4; ++q) hin[q] = live ? ld4_nt(reinterpret_cast<const float*>(hp + 64 * q)) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      float hv[16];
      m2_gemm(hv, xt, w1, inv_x, r, h);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        hv[4 * q] *= hin[q].x;
        hv[4 * q + 1] *= hin[q].y;
        hv[4 * q + 2] *= hin[q].z;
        hv[4 * q + 3] *= hin[q].w;
      }
      pmax[(nb * 2 + h) * 32 + r] = m2_amax16(hv);
      __syncthreads();
      const float inv_h = m2_commit(hv, ht, pmax, nb, r, h);
      __syncthreads();
      float4 add[4];
      if (mlp == 1) {
        const float4* ap = reinterpret_cast<const float4*>(A.g_msg + pg * NF + nb * 32 + 4 * h);
#pragma unroll
        for (int q = 0; q < 4; ++q) add[q] = live ? ap[2 * q] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      float y[16];
      m2_gemm(y, ht, w2, inv_h, r, h);
      if (mlp == 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          y[4 * q] += add[q].x;
          y[4 * q + 1] += add[q].y;
          y[4 * q + 2] += add[q].z;
          y[4 * q + 3] += add[q].w;
        }
      }
      if (t + 1 < nT) load_x(t + 1);
      if (!last) {
        if (live) {
          float4* yp = reinterpret_cast<float4*>(A.g_msg + pg * NF + nb * 32 + 4 * h);
#pragma unroll
          for (int q = 0; q < 4; ++q) yp[2 * q] = make_float4(y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]);
        }
      } else {
        // ---- B3: the message adjoint of this tile
        __syncthreads();                                  // every wave is done reading the hidden tile: g_msg takes its place
        {
          float* yr = phit + r * M2_PHI_PITCH + nb * 32 + 4 * h;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(yr + 8 * q) = make_float4(y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]);
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int pq = 32 * t + 8 * u + 2 * nb + h;
          if (pq < nP) {
            const int ij = sm_pij[pq], i = ij & 255, j = ij >> 8;
            const int2 gx = sm_xg[pq];
            const FilterW fw = filter_weights(__int_as_float(gx.y));
            float4 eps, deps;
            filter_value_deriv(A.table, gx.x, c4, fw, eps, deps);
            float* row = phit + (pq - 32 * t) * M2_PHI_PITCH + c4;
            const float4 G = add4(add4(*reinterpret_cast<const float4*>(row), *reinterpret_cast<const float4*>(sm_ga + i * NF + c4)),
                                  *reinterpret_cast<const float4*>(sm_ga + j * NF + c4));
            const float4 mi = *reinterpret_cast<const float4*>(sm_m + i * NF + c4), mj = *reinterpret_cast<const float4*>(sm_m + j * NF + c4);
            const float gxs = half_sum_top(dot4(mul4(mul4(G, mi), mj), deps));
            if (r == 31) {     // the owner's edge carries all of g_x (edge.hip:msg_bwd_kernel)
              const int e = owner_edge(pq, i);
              A.g_x[e] = gxs;
              A.g_x[A.rev[e]] = 0.f;
            }
            if (LOWER) *reinterpret_cast<float4*>(row) = mul4(G, eps);
          }
        }
        if (LOWER) {
          __syncthreads();
#pragma unroll
          for (int k = 0; k < M2_OWN; ++k) {
            const int a = nb + M2_WAVES * k;
            if (a < n) {
              const int eb = __builtin_amdgcn_readlane(rt_k[k], t), ee = __builtin_amdgcn_readlane(rt_k[k], t + 1);
              for (int l = eb; l < ee; l += M2_AGG) {
                float2 v[M2_AGG], mj[M2_AGG];
#pragma unroll
                for (int u = 0; u < M2_AGG; ++u) {
                  const bool on = l + u < ee;
                  const int iu = __builtin_amdgcn_readlane(inc_k[k], on ? l + u : l);
                  v[u] = *reinterpret_cast<const float2*>(phit + ((iu & 511) - 32 * t) * M2_PHI_PITCH + c2);
                  if (!on) v[u] = make_float2(0.f, 0.f);
                  mj[u] = *reinterpret_cast<const float2*>(sm_m + ((iu >> 9) & 31) * NF + c2);
                }
#pragma unroll
                for (int u = 0; u < M2_AGG; ++u) acc_m[k] = fma2(v[u], mj[u], acc_m[k]);
              }
            }
          }
        }
      }
    }
  };
  if (LOWER) {
    run_mlp(std::integral_constant<int, 0>(), std::false_type());
    __syncthreads();        // (the first term's rows are in the L2 before any wave reads one back)
    run_mlp(std::integral_constant<int, 1>(), std::true_type());
#pragma unroll
    for (int k = 0; k < M2_OWN; ++k) {
      const int a = nb + M2_WAVES * k;
      if (a < n) st2(A.g_m + (size_t)(a0 + a) * NF + c2, acc_m[k]);
    }
  } else {
    run_mlp(std::integral_constant<int, 0>(), std::true_type());
  }
  dbg.stamp();
  };
  // ---- the queue (as the forward)
  int it = 0;
  for (int q = blockIdx.x; q < A.n_mol; ++it) {
    int nxt = 0;
    if (tid0 == 0) nxt = (int)gridDim.x + atomicAdd(A.queue, 1);
    molecule(A.order[q]);
    if (tid0 == 0) sm_q[it & 1] = nxt;
    __syncthreads();
    q = sm_q[it & 1];
#ifdef MF_CLOCK_DEBUG
    if (it == 0) dbg.print(LOWER ? "mol2_bwd<1>" : "mol2_bwd<0>");
#endif
  }
}

int launch_mol2_edge_bwd(bool lower, const int* mol_ptr, const int* row_ptr, const int* pair_ptr, const int* col, const int* pid,
                         const int* rev, const float* geo, const int* xg, const float* gf, const float* g_a, const float* m,
                         const float* f_in, const float* table, const char* img12T, const char* img10T, const char* img22T,
                         const char* img20T, const float* h1, const float* h2, const float* phi1, const float* phi2, float* g_fin,
                         float* g_m, float* g_x, float* g_u, float* g_phi, float* g_msg, const int* order, int* queue, int n_mol,
                         hipStream_t s) {
  ScopedTimer t0(TC_MOL_BWD, s);
  static const hipError_t rc0 = hipFuncSetAttribute((const void*)mol2_edge_bwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, M2B_LDS);
  static const hipError_t rc1 = hipFuncSetAttribute((const void*)mol2_edge_bwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, M2B_LDS);
  HIP_TRY(rc0);
  HIP_TRY(rc1);
  if (n_mol <= 0) return 0;
  Mol2BwdArgs A = {mol_ptr, row_ptr, pair_ptr, col, pid, rev, geo, reinterpret_cast<const int2*>(xg), gf, g_a, m, f_in, table,
                   img12T, img10T, img22T, img20T, h1, h2, phi1, phi2, g_fin, g_m, g_x, g_u, g_phi, g_msg, order, queue, n_mol};
  const int grid = n_mol < M2_GRID ? n_mol : M2_GRID;
  if (lower)
    mol2_edge_bwd_kernel<true><<<grid, M2_THREADS, M2B_LDS, s>>>(A);
  else
    mol2_edge_bwd_kernel<false><<<grid, M2_THREADS, M2B_LDS, s>>>(A);
  LAUNCH_CHECK();
  return 0;
}
