// The two edge MLPs of a layer (equiv_message1 / equiv_message2, newtonnet/models/newtonnet.py:188-197,218,222) and their adjoint
// over the pair rows of a large batch, with the weights RESIDENT IN REGISTERS (gfx950, split-f16 products as in mlp128s.hip).
//
// Why another form of the same arithmetic.  mlp128s.hip keeps the two matrices of ONE MLP in LDS (2 x 68 KiB), so a launch runs the
// two MLPs as two phases: the forward reads the message rows twice, and the adjoint writes g_msg, reads it back and writes it again.
// These launches are bound by HBM, the store side in particular (plain streaming kernels on this pool: read 6.2, write 4.5-5.1
// TB/s, tools/ubench/hbm_stream.hip), so the second pass is paid in full.  Here an 8-wave workgroup holds ALL FOUR matrices of the
// layer as MFMA A fragments in registers -- wave (g, nb) keeps output block nb of both matrices of MLP g, 2 x 64 VGPRs, loaded once
// per launch from the prepared fragment-order images (node128s.hip:weight_image_kernel) -- and a 32-row tile passes through both
// MLPs at once:
//   forward : msg tile read ONCE -> h1 | h2 (kept for the adjoint, fragment order) -> phi1 | phi2          5 row passes instead of 6
//   adjoint : g_phi1 | g_phi2, h1 | h2 -> both terms of g_msg summed on chip -> g_msg written ONCE        5 row passes instead of 7
// Activations travel between the stages through LDS tiles in split-f16 fragment layout (17 KiB each); LDS holds no weights.
//
// Tile flow (group g = waves 4g .. 4g+3 = MLP g; one wave of each group per SIMD).  The groups are synchronised among their own
// four waves only, by arrival counters in LDS (rw_group_sync) -- s_barrier would hold both groups in lockstep, and then the MFMA
// phase of one never overlaps the VALU phase of the other on the SIMD they share (measured: 116 us in lockstep, 104 us apart):
//   1. every wave brings its share of the tile's input rows (requested one tile ahead, whole 512-byte rows per half-wave), scales
//      each row by its largest magnitude (a DPP reduction: the row lives in 32 lanes), splits it and writes the X tile
//      (adjoint: the group's own g_phi_g tile; forward: ONE msg tile for both groups, double-buffered, behind an 8-wave arrival)
//   2. stage 1: wave (g, nb) forms block nb of W1_g . X^T from its register fragments; the forward leaves silu'(h) in the scratch
//      the adjoint reads (mlp128s.hip) and applies SiLU, the adjoint multiplies by that silu'(h); the row maxima of the result go
//      through LDS (node128s.hip:tile_publish)
//   3. stage 2: the same wave forms block nb of W2_g . A_g^T; forward: phi_g rows; adjoint: wave (1, nb) hands its block to wave
//      (0, nb) through a double-buffered LDS slot (ready / taken counters), which adds and stores g_msg
// Products, scales and the order of accumulation are those of mlp128s.hip (hi hi + hi lo + lo hi per 16 k-values, rows scaled per
// row, matrices per matrix): the two forms agree to the last bit or two
// (tests/test_hip_parity.py::test_one_pass_edge_mlp_form_agrees_with_the_two_phase_form holds both against float64).
// The kernel is bound by VALU issue (~760 wave-instructions per wave and tile, 1.5x mlp128s.hip -- the price of passing
// activations through LDS instead of handing them over in registers), which is why only the adjoint launches take it by default.
#include <stdlib.h>

#include "common.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef __bf16 b4 __attribute__((ext_vector_type(4)));

#define RW_WAVES 8
#define RW_THREADS (64 * RW_WAVES)
#define RW_PITCH 272                       // bytes per row of one plane of an LDS tile (128 f16 + 16 pad: conflict-free b128 reads)
#define RW_PLANE (32 * RW_PITCH)
#define RW_TILE (2 * RW_PLANE)             // hi plane, lo plane
#define RW_WIMG_PLANE (NF * NF * 2)

struct RwLds {
  char xt[2][RW_TILE];          // stage-1 operand tiles, one per group (forward: both hold the msg tile; adjoint: g_phi1 / g_phi2)
  char at[2][RW_TILE];          // stage-2 operand tiles, one per MLP
  float pmax[2][8 * 32];        // per (wave of the group, lane half) maxima of each row of the stage-1 result
  float invx[2][32];            // inverse row scales of the X tiles
  float4 part[2][4][4 * 64];    // adjoint: group 1's stage-2 blocks on their way to group 0, double-buffered
  unsigned bar[2];              // arrival counters of the two 4-wave groups (rw_group_sync)
  unsigned xbar;                // forward: arrivals of all 8 waves at the shared X tile
  unsigned ready[4], taken[4];  // hand-over counters per block: tiles published by wave (1, nb) / consumed by wave (0, nb)
};

struct RwFrag {
  h8 hi[8], lo[8];
  float inv;
};

__device__ __forceinline__ void rw_pow2_scale(float m, float& S, float& inv) {
  const int e = (int)((__float_as_uint(m) >> 23) & 0xffu);
  const bool ok = e >= 40 && e < 255;
  S = ok ? __uint_as_float((unsigned)(268 - e) << 23) : 1.0f;
  inv = ok ? __uint_as_float((unsigned)(e - 14) << 23) : 1.0f;
}
// largest value of each 32-lane half (inputs >= 0), in every lane of the half: DPP folds + two v_readlane (common.h:half_sum_top)
#define RW_DPP_MAX(v, ctrl, row_mask) \
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, row_mask, 0xF, false)))
__device__ __forceinline__ float rw_half_max(float v, int h) {
  RW_DPP_MAX(v, 0xB1, 0xF);
  RW_DPP_MAX(v, 0x4E, 0xF);
  RW_DPP_MAX(v, 0x141, 0xF);
  RW_DPP_MAX(v, 0x140, 0xF);
  RW_DPP_MAX(v, 0x142, 0xA);
  const float lo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31));
  const float hi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
  return h ? hi : lo;
}
// Barrier of ONE 4-wave group (gfx950 has no named barriers; s_barrier would hold both groups in lockstep, and then the MFMA
// phase of one never overlaps the VALU phase of the other on the SIMD they share).  A counter in LDS with workgroup-scope
// release / acquire fences restricted to the LDS address space (`__builtin_amdgcn_fence(order, "workgroup", "local")`: the
// compiler orders this wave's tile accesses against the counter and emits s_waitcnt lgkmcnt(0) only) -- no vmcnt wait, the
// prefetched rows and the streaming stores stay in flight across it.  The spin is bounded (RW_SPIN_LIMIT polls, several seconds:
// far beyond anything a profiler's serialisation can add) so that a lost arrival aborts the launch instead of hanging the GPU.
#ifndef RW_SPIN_LIMIT
#define RW_SPIN_LIMIT (1u << 26)
#endif
__device__ __forceinline__ void rw_release_lds() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); }
__device__ __forceinline__ void rw_acquire_lds() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); }
__device__ __forceinline__ void rw_spin_until(const unsigned* cnt, unsigned target) {
  unsigned spins = 0;
  while ((int)(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - target) < 0) {
    if (++spins > RW_SPIN_LIMIT) __builtin_trap();   // seconds without the partner: abort the launch (a HIP error), never a silent result
    __builtin_amdgcn_s_sleep(1);
  }
  rw_acquire_lds();
}
__device__ __forceinline__ void rw_signal(unsigned* cnt, int lane) {
  rw_release_lds();
  if (lane == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void rw_group_sync(unsigned* cnt, unsigned& target, int lane, unsigned waves = 4) {
  target += waves;
  rw_signal(cnt, lane);
  rw_spin_until(cnt, target);
}
// A fragments of output block nb from a fragment-order image (node128s.hip:load_wimg)
__device__ __forceinline__ void rw_load_frag(RwFrag& w, const char* __restrict__ img, int nb, int r, int h) {
  const char* p = img + ((size_t)(nb * 8 * 64 + h * 32 + r) << 4);
#pragma unroll
  for (int T = 0; T < 8; ++T) {
    w.hi[T] = *reinterpret_cast<const h8*>(p + 1024 * T);
    w.lo[T] = *reinterpret_cast<const h8*>(p + RW_WIMG_PLANE + 1024 * T);
  }
  w.inv = *reinterpret_cast<const float*>(img + 2 * RW_WIMG_PLANE);
}
// block of D^T = W . X^T for the 32 rows of an LDS tile (node128s.hip:tile_gemm_s)
__device__ __forceinline__ f32x16 rw_gemm(const char* tile, const RwFrag& w, int r, int h) {
  f32x16 acc;
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.f;
  const char* xr = tile + r * RW_PITCH + 16 * h;
  h8 bh0 = *reinterpret_cast<const h8*>(xr), bl0 = *reinterpret_cast<const h8*>(xr + RW_PLANE);
  h8 bh1 = *reinterpret_cast<const h8*>(xr + 32), bl1 = *reinterpret_cast<const h8*>(xr + RW_PLANE + 32);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int T = 0; T < 8; ++T) {
    h8 bh2, bl2;
    if (T < 6) {
      bh2 = *reinterpret_cast<const h8*>(xr + 32 * (T + 2));
      bl2 = *reinterpret_cast<const h8*>(xr + RW_PLANE + 32 * (T + 2));
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
  return acc;
}
// bf16 compute mode (training under autocast(bfloat16); the images are WIMG_FMT_BF16): one bf16 plane per operand, no scales, one
// v_mfma_f32_32x32x16_bf16 per 16 k-values
__device__ __forceinline__ f32x16 rw_gemm_bf(const char* tile, const RwFrag& w, int r, int h) {
  f32x16 acc;
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.f;
  const char* xr = tile + r * RW_PITCH + 16 * h;
  b8 b0 = *reinterpret_cast<const b8*>(xr), b1 = *reinterpret_cast<const b8*>(xr + 32);
#pragma unroll
  for (int T = 0; T < 8; ++T) {
    b8 b2;
    if (T < 6) b2 = *reinterpret_cast<const b8*>(xr + 32 * (T + 2));
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, w.hi[T]), b0, acc, 0, 0, 0);
    b0 = b1;
    if (T < 6) b1 = b2;
  }
  return acc;
}
__device__ __forceinline__ void rw_commit_row_bf(const float4& v, char* tile, float* invx, int row, int c) {
  b4 w;
  w[0] = (__bf16)v.x, w[1] = (__bf16)v.y, w[2] = (__bf16)v.z, w[3] = (__bf16)v.w;
  *reinterpret_cast<b4*>(tile + row * RW_PITCH + 2 * ((c >> 2) * 16 + (c & 1) * 8 + ((c >> 1) & 1) * 4)) = w;
  if (c == 0) invx[row] = 1.0f;
}
__device__ __forceinline__ void rw_commit_block_bf(const float (&v)[16], char* tile, int nb, int r, int h) {
  char* row = tile + r * RW_PITCH + 2 * (nb * 32 + 8 * h);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    b4 w;
#pragma unroll
    for (int c = 0; c < 4; ++c) w[c] = (__bf16)v[4 * q + c];
    *reinterpret_cast<b4*>(row + 2 * ((q >> 1) * 16 + (q & 1) * 4)) = w;
  }
}
// one 512-byte input row per half-wave: scale by the row maximum, split, write into an X tile (k-slot 16 T + 8 h' + 4 j + c <->
// feature 16 T + 8 j + 4 h' + c, the permutation of the images); lane & 31 == 0 keeps the inverse scale
__device__ __forceinline__ void rw_commit_row(const float4& v, char* tile, float* invx, int row, int c, int h) {
  const float m = rw_half_max(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))), h);
  float S, inv;
  rw_pow2_scale(m, S, inv);
  const float s[4] = {v.x * S, v.y * S, v.z * S, v.w * S};
  h4 hi, lo;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const _Float16 a = (_Float16)s[j];
    hi[j] = a;
    lo[j] = (_Float16)(s[j] - (float)a);
  }
  char* p = tile + row * RW_PITCH + 2 * ((c >> 2) * 16 + (c & 1) * 8 + ((c >> 1) & 1) * 4);
  *reinterpret_cast<h4*>(p) = hi;
  *reinterpret_cast<h4*>(p + RW_PLANE) = lo;
  if (c == 0) invx[row] = inv;
}
// stage-1 result of this lane (16 values of block nb) -> the group's stage-2 tile, scaled by the row maximum (node128s.hip:tile_commit)
__device__ __forceinline__ float rw_commit_block(const float (&v)[16], char* tile, const float* pmax, int nb, int r, int h) {
  float m = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) m = fmaxf(m, pmax[j * 32 + r]);
  float S, inv;
  rw_pow2_scale(m, S, inv);
  char* row = tile + r * RW_PITCH + 2 * (nb * 32 + 8 * h);
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
    *reinterpret_cast<h4*>(row + RW_PLANE + off) = lo;
  }
  return inv;
}

// MODE_FWD: a[0] / a[1] = the two MLPs over the same X (msg);  MODE_BWD: a[g].X = g_phi_g, a[g].H = h_g, both Y = g_msg.
// H is in fragment order (h_frag) in those two modes; W1_img / W2_img are the stage-1 / stage-2 images of each MLP.
// MODE_TAN / MODE_TAN2 (round 6): the adjoint-shaped launches of the TRAINING sweeps in the same one-pass form -- the value adjoint
// that keeps its stage-1 product (T = g_phi V2, row-major, for the tangent sweeps and the weight gradients) and the tangent of that
// adjoint (G = dT act'(h) + T2 act''(h) Hd, kept row-major).  There H holds the pre-activations themselves, row-major (the training
// path needs act, act' and act'' of them), so the epilogue evaluates silu' / silu'' instead of reading a kept factor; both terms of
// g_msg / dg_msg are summed on chip and written ONCE (the two-phase form of mlp128s.hip wrote, re-read and re-wrote them: 9 instead
// of 11 row passes for MODE_TAN, 13 instead of 15 for MODE_TAN2).
#ifdef RW_CLOCK_DEBUG   // tooling: wall-clock (100 MHz) time of every phase of the tile loop, summed over the tiles of one wave
#define RW_DBG_DECL() long long dbg_t[12]; long long dbg_s[12]; int dbg_k = 0; int dbg_tiles = 0; for (int k_ = 0; k_ < 12; ++k_) dbg_s[k_] = 0;
#define RW_DBG_TOP() dbg_k = 0; dbg_t[0] = wall_clock64(); ++dbg_tiles;
#define RW_DBG() { ++dbg_k; dbg_t[dbg_k] = wall_clock64(); dbg_s[dbg_k] += dbg_t[dbg_k] - dbg_t[dbg_k - 1]; }
#define RW_DBG_PRINT()                                                                                                        \
  if (blockIdx.x == 3 && lane == 0 && (wave == 0 || wave == 5)) {                                                              \
    const double d_ = 100.0 * dbg_tiles;                                                                                      \
    printf("regw mode %d wave %d: %d tiles, us per tile: waitX %.2f commitX %.2f req+bar %.2f gemm1 %.2f epi1 %.2f bar %.2f commit+bar %.2f gemm2 %.2f epi2 %.2f\n", \
           MODE, wave, dbg_tiles, dbg_s[1] / d_, dbg_s[2] / d_, dbg_s[3] / d_, dbg_s[4] / d_, dbg_s[5] / d_, dbg_s[6] / d_, dbg_s[7] / d_,        \
           dbg_s[8] / d_, dbg_s[9] / d_);                                                                                                    \
  }
#else
#define RW_DBG_DECL()
#define RW_DBG_TOP()
#define RW_DBG()
#define RW_DBG_PRINT()
#endif

// SINGLE (adjoint only): ONE MLP (layer 0 has no equiv_message2 term) -- both groups hold the same two matrices and take
// alternate tiles, each storing its own g_msg rows; the groups then never meet (no hand-over, own trip counts).
// SHARED_X: the two MLPs read the SAME input rows and write their own outputs (the forward; the tangent forward of training, MODE_TAN
// with X = dmsg); otherwise every group has its own input and the two stage-2 results are summed into one output (the adjoints).
template <int MODE, bool SINGLE = false, bool SHARED_X = (MODE == MODE_FWD && !SINGLE)>
__global__ void __launch_bounds__(RW_THREADS, 2) mlp_regw_kernel(const MlpPair P) {
  constexpr bool OWN_X = !SHARED_X;                    // every group stages its own X tile (a shared-X launch stages one for both)
  constexpr bool TRAIN = MODE == MODE_TAN || MODE == MODE_TAN2;   // H = pre-activations, row-major; T / G kept row-major
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  RwLds& L = *reinterpret_cast<RwLds*>(lds_raw);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = wave >> 2, nb = wave & 3;
  const int r = lane & 31, h = lane >> 5;
  const int M = mlp_rows(P.a[0]);
  if (M <= 0) return;   // (uniform; only possible with a device-side count)
  const int n_tiles = (M + 31) >> 5;
  // (uniform) the bf16 compute mode of the training sweeps: what the images say (node128s.hip: nnhip_weight_images_bf16)
  const bool bf = TRAIN && *reinterpret_cast<const int*>(P.a[0].W1_img + 2 * RW_WIMG_PLANE + 4) == WIMG_FMT_BF16;

  // this wave's weights: block nb of both matrices of MLP g, for the whole launch
  const int ga = SINGLE ? 0 : g;    // whose arguments and matrices this wave works with
  RwFrag w1, w2;
  rw_load_frag(w1, ga ? P.a[1].W1_img : P.a[0].W1_img, nb, r, h);
  rw_load_frag(w2, ga ? P.a[1].W2_img : P.a[0].W2_img, nb, r, h);

  // input rows this wave brings per tile.  Forward: 4 of the 32 msg rows -- ONE X tile serves both groups, double-buffered by tile
  // parity (a wave writes tile k + 1 only after its own tile k, and nobody passes the 8-wave arrival of tile k before every wave
  // has left tile k - 1, so the buffer of tile k - 1 is free by then).  Adjoint: 8 of the 32 rows of the group's own g_phi_g.
  constexpr int NX = OWN_X ? 4 : 2;
  const float* Xg = (SHARED_X || ga == 0) ? P.a[0].X : P.a[1].X;
  const int ldx = (SHARED_X || ga == 0) ? P.a[0].ldx : P.a[1].ldx;
  const int xrow0 = OWN_X ? 8 * nb : 4 * wave;
  const float* Hg = ga ? P.a[1].H : P.a[0].H;
  float* Yg = ga ? P.a[1].Y : P.a[0].Y;
  const int ldy = ga ? P.a[1].ldy : P.a[0].ldy;
  const int ldh = ga ? P.a[1].ldh : P.a[0].ldh;                 // (training modes: row pitch of H / T / T2 / Hd / G)
  float* Tg = ga ? P.a[1].T : P.a[0].T;                         // MODE_TAN: out
  const float* T2g = ga ? P.a[1].T2 : P.a[0].T2;                // MODE_TAN2: in
  const float* Hdg = ga ? P.a[1].Hd : P.a[0].Hd;
  float* Gg = ga ? P.a[1].G : P.a[0].G;
  const int tile0 = SINGLE ? 2 * (int)blockIdx.x + g : (int)blockIdx.x;
  const int tile_step = SINGLE ? 2 * (int)gridDim.x : (int)gridDim.x;
  unsigned* bar = &L.bar[g];
  unsigned bar_target = 0, xbar_target = 0;

  if (threadIdx.x == 0) L.xbar = 0;
  if (threadIdx.x < 2) L.bar[threadIdx.x] = 0;
  if (threadIdx.x < 4) L.ready[threadIdx.x] = L.taken[threadIdx.x] = 0;
  __syncthreads();                  // (the only workgroup-wide barrier of the launch)

  float4 xq[NX], hq[4];
  auto request = [&](int tile) {
    const int tc = min(tile, n_tiles - 1);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int row = min((tc << 5) + xrow0 + 2 * i + h, M - 1);
      // (adjoint: g_phi is streamed once -- keep it from displacing the g_msg rows this launch writes for the next kernel)
      xq[i] = MODE == MODE_BWD ? ld4_nt(Xg + (size_t)row * ldx + 4 * r) : ld4(Xg + (size_t)row * ldx + 4 * r);
    }
    if (MODE == MODE_BWD) {
      const float4* hp = reinterpret_cast<const float4*>(Hg) + ((size_t)tc * 4 + nb) * 256 + lane;
#pragma unroll
      for (int q = 0; q < 4; ++q) hq[q] = ld4_nt(reinterpret_cast<const float*>(hp + 64 * q));
    }
    if (TRAIN) {   // this lane's 16 pre-activations of row r of the tile: features nb*32 + 8 q + 4 h + {0..3}
      const float* hp = Hg + (size_t)min((tc << 5) + r, M - 1) * ldh + nb * 32 + 4 * h;
#pragma unroll
      for (int q = 0; q < 4; ++q) hq[q] = ld4(hp + 8 * q);
    }
  };
  request(tile0);
  RW_DBG_DECL()

  int k = 0;                        // tiles done by this workgroup (SINGLE: by this group)
  for (int tile = tile0; tile < n_tiles; tile += tile_step, ++k) {
    RW_DBG_TOP()
    const int e = (tile << 5) + r;
    const bool live = e < M;
    char* xtile = L.xt[OWN_X ? g : (k & 1)];
    float* invx = L.invx[OWN_X ? g : (k & 1)];
    // ---------------- 1. the X tile (forward: shared; adjoint: the group's own)
#ifdef RW_CLOCK_DEBUG
    asm volatile("" ::"v"(xq[NX - 1].w));   // (the rows have arrived: phase 1 is the wait, phase 2 the commit)
    RW_DBG()
#endif
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      if (bf)
        rw_commit_row_bf(xq[i], xtile, invx, xrow0 + 2 * i + h, r);
      else
        rw_commit_row(xq[i], xtile, invx, xrow0 + 2 * i + h, r, h);
    }
    float4 hin[4];
    if (MODE != MODE_FWD) {
#pragma unroll
      for (int q = 0; q < 4; ++q) hin[q] = hq[q];
    }
    RW_DBG()   // 1: X commit (waits for the prefetched rows)
    request(tile + tile_step);      // the next tile's rows travel while this one is computed
    if (!OWN_X)
      rw_group_sync(&L.xbar, xbar_target, lane, 8);
    else
      rw_group_sync(bar, bar_target, lane);
    RW_DBG()   // 2: request + barrier
    // ---------------- 2. stage 1
    float a[16];
    {
      const f32x16 acc = bf ? rw_gemm_bf(xtile, w1, r, h) : rw_gemm(xtile, w1, r, h);
      RW_DBG()   // 3: GEMM 1
      const float sc = invx[r] * w1.inv;
      if (MODE == MODE_FWD) {
        float4* hp = reinterpret_cast<float4*>(ga ? P.a[1].H : P.a[0].H) + ((size_t)tile * 4 + nb) * 256 + lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 hv = make_float4(acc[4 * q] * sc, acc[4 * q + 1] * sc, acc[4 * q + 2] * sc, acc[4 * q + 3] * sc);
          // (the scratch keeps silu'(h), mlp128s.hip; the region is padded to whole tiles)
          st4_nt(reinterpret_cast<float*>(hp + 64 * q), make_float4(dsilu_f(hv.x), dsilu_f(hv.y), dsilu_f(hv.z), dsilu_f(hv.w)));
          a[4 * q] = silu_f(hv.x);
          a[4 * q + 1] = silu_f(hv.y);
          a[4 * q + 2] = silu_f(hv.z);
          a[4 * q + 3] = silu_f(hv.w);
        }
      } else if (TRAIN) {
        const size_t roff = (size_t)min(e, M - 1) * ldh + nb * 32 + 4 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 t = make_float4(acc[4 * q] * sc, acc[4 * q + 1] * sc, acc[4 * q + 2] * sc, acc[4 * q + 3] * sc);
          float4 v;
          if (MODE == MODE_TAN) {      // keep the stage-1 product, then the act' factor
            if (live) st4(Tg + (size_t)e * ldh + nb * 32 + 4 * h + 8 * q, t);
            v = make_float4(t.x * dsilu_f(hin[q].x), t.y * dsilu_f(hin[q].y), t.z * dsilu_f(hin[q].z), t.w * dsilu_f(hin[q].w));
          } else {                     // G = dT act'(h) + T2 act''(h) Hd, kept for the weight-gradient products
            const float4 t2 = ld4(T2g + roff + 8 * q), hd = ld4(Hdg + roff + 8 * q);
            v.x = fmaf(t.x, dsilu_f(hin[q].x), t2.x * d2silu_f(hin[q].x) * hd.x);
            v.y = fmaf(t.y, dsilu_f(hin[q].y), t2.y * d2silu_f(hin[q].y) * hd.y);
            v.z = fmaf(t.z, dsilu_f(hin[q].z), t2.z * d2silu_f(hin[q].z) * hd.z);
            v.w = fmaf(t.w, dsilu_f(hin[q].w), t2.w * d2silu_f(hin[q].w) * hd.w);
            if (live) st4(Gg + (size_t)e * ldh + nb * 32 + 4 * h + 8 * q, v);
          }
          a[4 * q] = v.x;
          a[4 * q + 1] = v.y;
          a[4 * q + 2] = v.z;
          a[4 * q + 3] = v.w;
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          a[4 * q] = acc[4 * q] * sc * hin[q].x;      // (hin = silu'(h), left by the forward)
          a[4 * q + 1] = acc[4 * q + 1] * sc * hin[q].y;
          a[4 * q + 2] = acc[4 * q + 2] * sc * hin[q].z;
          a[4 * q + 3] = acc[4 * q + 3] * sc * hin[q].w;
        }
      }
    }
    {
      float m = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) m = fmaxf(m, fabsf(a[q]));
      L.pmax[g][(nb * 2 + h) * 32 + r] = m;
    }
    RW_DBG()   // 4: stage-1 epilogue (H stores / loads, activation, publish)
    rw_group_sync(bar, bar_target, lane);   // the row maxima are visible; every wave of the group is done with the X tile
    RW_DBG()   // 5: barrier
    float inv2 = 1.0f;
    if (bf)
      rw_commit_block_bf(a, L.at[g], nb, r, h);
    else
      inv2 = rw_commit_block(a, L.at[g], L.pmax[g], nb, r, h);
    rw_group_sync(bar, bar_target, lane);
    RW_DBG()   // 6: commit + barrier
    // ---------------- 3. stage 2
    {
      const f32x16 acc = bf ? rw_gemm_bf(L.at[g], w2, r, h) : rw_gemm(L.at[g], w2, r, h);
      RW_DBG()   // 7: GEMM 2
      const float sc = inv2 * w2.inv;
      if (SHARED_X || SINGLE) {
        if (live) {
          float4* yp = reinterpret_cast<float4*>(Yg + (size_t)e * ldy + nb * 32 + 4 * h);
#pragma unroll
          for (int q = 0; q < 4; ++q)
            yp[2 * q] = make_float4(acc[4 * q] * sc, acc[4 * q + 1] * sc, acc[4 * q + 2] * sc, acc[4 * q + 3] * sc);
        }
      } else if (g == 1) {
        // wave (1, nb) -> wave (0, nb): buffer k & 1 is free once tile k - 2 has been taken
        rw_spin_until(&L.taken[nb], (unsigned)(k - 1));
#pragma unroll
        for (int q = 0; q < 4; ++q)
          L.part[k & 1][nb][q * 64 + lane] = make_float4(acc[4 * q] * sc, acc[4 * q + 1] * sc, acc[4 * q + 2] * sc, acc[4 * q + 3] * sc);
        rw_signal(&L.ready[nb], lane);
      } else {
        rw_spin_until(&L.ready[nb], (unsigned)(k + 1));
        float4 o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = L.part[k & 1][nb][q * 64 + lane];
        rw_signal(&L.taken[nb], lane);
        if (live) {
          float4* yp = reinterpret_cast<float4*>(Yg + (size_t)e * ldy + nb * 32 + 4 * h);
#pragma unroll
          for (int q = 0; q < 4; ++q)
            yp[2 * q] = make_float4(acc[4 * q] * sc + o[q].x, acc[4 * q + 1] * sc + o[q].y, acc[4 * q + 2] * sc + o[q].z,
                                    acc[4 * q + 3] * sc + o[q].w);
        }
      }
    }
    RW_DBG()   // 8: stage-2 epilogue (stores; adjoint: the hand-over)
  }
  RW_DBG_PRINT()
}

// NNHIP_MLP_REGW: 1 (default) = the adjoint launches take this form (measured 106 us against 121 us per launch on the config-2
// batch: g_msg is written once); 2 = the forward launches too (108 us against 106 us: the message rows are read once, but the
// per-tile cost of passing activations through LDS eats the gain -- mlp128s.hip hands them over in registers); 0 = off
static int mlp_regw_level() {
  static const int level = [] {
    const char* v = getenv("NNHIP_MLP_REGW");
    return v ? atoi(v) : 1;
  }();
  return level;
}
// one MLP (layer 0): NNHIP_MLP_REGW_SINGLE = 0 off, 1 the adjoint, 2 the forward too (A/B timing)
static int mlp_regw_single() {
  static const int single = getenv("NNHIP_MLP_REGW_SINGLE") ? atoi(getenv("NNHIP_MLP_REGW_SINGLE")) : 1;
  return single;
}
// what the library does with the switches above, for callers that model its traffic (bench.py); include/newtonnet_hip.h
extern "C" int nnhip_mlp_forms(void) {
  const bool split = split_products_enabled();
  const int level = split ? mlp_regw_level() : 0, single = level > 0 ? mlp_regw_single() : 0;
  return (split ? 1 : 0) | (level >= 1 ? 2 : 0) | (level >= 2 ? 4 : 0) | (single >= 1 ? 8 : 0) | (single >= 2 ? 16 : 0);
}
bool mlp_regw_serves(int mode, const MlpPair& P) {
  // (the bf16 compute mode exists in the training modes only: the kernel reads it off the weight images)
  if ((P.a[0].bf16 || (P.n > 1 && P.a[1].bf16)) && mode != MODE_TAN && mode != MODE_TAN2) return false;
  const int level = mlp_regw_level();
  const int single = mlp_regw_single();
  if (mode == MODE_TAN || mode == MODE_TAN2) {
    // the adjoint-shaped launches of the training sweeps: one MLP (layer 0, no accumulate), or two MLPs whose outputs are summed
    // into one Y (the second accumulating); H / T / T2 / Hd / G row-major with one pitch.  NNHIP_MLP_REGW_TRAIN=0: the two-phase form
    static const bool train_on = !(getenv("NNHIP_MLP_REGW_TRAIN") && atoi(getenv("NNHIP_MLP_REGW_TRAIN")) == 0);
    if (level <= 0 || !train_on || P.n < 1 || P.n > 2) return false;
    for (int k = 0; k < P.n; ++k) {
      const MlpArgs& a = P.a[k];
      if (!a.W1_img || !a.W2_img || a.h_frag || a.act != NNHIP_ACT_SILU || a.b1 || a.b2 || a.M_dev || a.ldh != P.a[0].ldh) return false;
      if (mode == MODE_TAN ? !a.T : (!a.T2 || !a.Hd || !a.G)) return false;
    }
    if (P.n == 1) return !P.accum[0];
    if (mode == MODE_TAN && P.a[0].X == P.a[1].X && P.a[0].ldx == P.a[1].ldx && P.a[0].Y != P.a[1].Y && !P.accum[0] && !P.accum[1])
      return true;      // the tangent forward: one input (dmsg), two outputs -- the shared-X shape of the forward
    return P.a[0].Y == P.a[1].Y && P.a[0].ldy == P.a[1].ldy && !P.accum[0] && P.accum[1];
  }
  if (P.n == 1)
    return level > 0 && ((mode == MODE_BWD && single >= 1) || (mode == MODE_FWD && single >= 2)) && P.a[0].W1_img &&
           P.a[0].W2_img && P.a[0].h_frag && P.a[0].act == NNHIP_ACT_SILU && !P.a[0].b1 && !P.a[0].b2 && !P.accum[0];
  if (level <= 0 || P.n != 2 || !((mode == MODE_BWD) || (mode == MODE_FWD && level >= 2))) return false;
  for (int k = 0; k < 2; ++k)
    if (!P.a[k].W1_img || !P.a[k].W2_img || !P.a[k].h_frag || P.a[k].act != NNHIP_ACT_SILU || P.a[k].b1 || P.a[k].b2) return false;
  if (mode == MODE_FWD) return P.a[0].X == P.a[1].X && P.a[0].ldx == P.a[1].ldx && !P.accum[0] && !P.accum[1];
  return P.a[0].Y == P.a[1].Y && P.a[0].ldy == P.a[1].ldy && !P.accum[0] && P.accum[1];
}
template <int MODE, bool SINGLE, bool SHARED_X = (MODE == MODE_FWD && !SINGLE)>
static int launch_regw_t(const MlpPair& P, hipStream_t s) {
  static const hipError_t attr_rc = hipFuncSetAttribute((const void*)mlp_regw_kernel<MODE, SINGLE, SHARED_X>,
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(RwLds));
  HIP_TRY(attr_rc);
  const int n_tiles = (P.a[0].M + 31) / 32;
  const int units = SINGLE ? (n_tiles + 1) / 2 : n_tiles;   // (SINGLE: a workgroup takes two tiles per trip, one per group)
  const int blocks = units < 256 ? units : 256;              // one persistent workgroup per CU
  mlp_regw_kernel<MODE, SINGLE, SHARED_X><<<blocks, RW_THREADS, sizeof(RwLds), s>>>(P);
  LAUNCH_CHECK();
  return 0;
}
int launch_mlp_regw(int mode, const MlpPair& P, hipStream_t s) {
  if (mode == MODE_TAN && P.n == 2 && P.a[0].X == P.a[1].X && !P.accum[1]) return launch_regw_t<MODE_TAN, false, true>(P, s);
  if (mode == MODE_TAN) return P.n == 1 ? launch_regw_t<MODE_TAN, true>(P, s) : launch_regw_t<MODE_TAN, false>(P, s);
  if (mode == MODE_TAN2) return P.n == 1 ? launch_regw_t<MODE_TAN2, true>(P, s) : launch_regw_t<MODE_TAN2, false>(P, s);
  if (P.n == 1) return mode == MODE_FWD ? launch_regw_t<MODE_FWD, true>(P, s) : launch_regw_t<MODE_BWD, true>(P, s);
  return mode == MODE_FWD ? launch_regw_t<MODE_FWD, false>(P, s) : launch_regw_t<MODE_BWD, false>(P, s);
}
