// EXPERIMENTS, not in the build (round 4): two more forms of force_fwd for batches of small molecules, both bit-checked against the
// oracle at config-2 size through the product path (NNHIP_FORCE_FWD_MOL=2 / 3 selected them in edge.hip:launch_force_fwd) and both
// SLOWER than force_fwd_mol_kernel (edge_force_fwd class per step, same box):
//   force_fwd_pairs_kernel  every pair row read from memory once, streamed through LDS in chunks of 32 rows   0.171 vs 0.159 ms
//                           (profiles/r04_force_fwd_pairs_once_ab.txt; 80 KB of LDS = two 8-wave workgroups per CU in lock-step
//                           phases: the memory pipe idles between the phases)
//   force_fwd_half_kernel   a half-wave per row, all rows of a molecule in one round                            0.164 vs 0.158 ms
//                           (profiles/r04_force_fwd_half_wave_ab.txt)
// To try them again: paste the kernels into edge.hip in front of "adjoint of force_fwd" and the launcher lines at the bottom of this
// file into launch_force_fwd in front of the force_fwd_mol_kernel branch.

// ---------------------------------------------------------------------------------------------
// force_fwd for batches of small molecules, every PAIR ROW read from memory ONCE per launch: the molecule's pair rows
// [pair_ptr[a0], pair_ptr[a0 + n]) are contiguous, and the workgroup streams them through LDS in chunks of FP_PAIRS rows (one
// coalesced pass; the next chunk's loads fly under this chunk's arithmetic); both endpoints of a pair take the row from LDS.
// force_fwd_mol_kernel runs at the fabric's read ceiling (355 MB per launch in 59 us = 6 TB/s, profiles/r04_v3_*) and 160 of
// those MB are the SECOND read of every phi row -- 128 workgroups per XCD stream 19 MB between the two reads of a row, the L2
// holds 4.  The molecule's f_in rows and per-edge scalars (geo, local pair index, local sender) are staged once; a wave keeps
// the sums of its (at most three) rows in registers across the chunks and walks two cursors per row -- the pairs the other
// endpoint owns and the pairs it owns, both ascending in pair index.  Lane = 2 features (64 lanes x 8 B: one edge per LDS read).
// HAS_F = false: layer 0 (f_in = 0: the phi2 term vanishes, nothing but phi1 and geo is read).
// ---------------------------------------------------------------------------------------------
#define FP_PAIRS 32
#define FP_THREADS 512
#define FP_MAX_EDGES (NNHIP_MOL_STAGE_MAX * (NNHIP_MOL_STAGE_MAX - 1))
#define FP_ROWS_PER_WAVE ((NNHIP_MOL_STAGE_MAX + FP_THREADS / 64 - 1) / (FP_THREADS / 64))
// one row's edges whose pair rows are in the chunk [p_lo, p_hi) now in `buf`: first the pairs the other endpoint owns, then its own
template <bool HAS_F>
__device__ __forceinline__ void fp_row_chunk(int& elo, const int emid, int& ehi, const int eend, float2& s0, float2& s1, float2& s2,
                                             const int p_lo, const int p_hi, const int lane, const float* __restrict__ buf,
                                             const float* __restrict__ fl, const float4* __restrict__ sgeo,
                                             const unsigned short* __restrict__ spid, const unsigned char* __restrict__ scol) {
  auto edge = [&](const int u) {
    const int cj = scol[u];
    if (cj & 0x80) return;
    const int pl = (int)spid[u] - p_lo;
    const float4 g = sgeo[u];
    const float2 v1 = *reinterpret_cast<const float2*>(buf + pl * NF + 2 * lane);
    s0.x = fmaf(v1.x, g.x, s0.x), s0.y = fmaf(v1.y, g.x, s0.y);
    s1.x = fmaf(v1.x, g.y, s1.x), s1.y = fmaf(v1.y, g.y, s1.y);
    s2.x = fmaf(v1.x, g.z, s2.x), s2.y = fmaf(v1.y, g.z, s2.y);
    if (HAS_F) {
      const float2 v2 = *reinterpret_cast<const float2*>(buf + (FP_PAIRS + pl) * NF + 2 * lane);
      const float2 f0 = *reinterpret_cast<const float2*>(fl + (cj * 3 + 0) * NF + 2 * lane);
      const float2 f1 = *reinterpret_cast<const float2*>(fl + (cj * 3 + 1) * NF + 2 * lane);
      const float2 f2 = *reinterpret_cast<const float2*>(fl + (cj * 3 + 2) * NF + 2 * lane);
      s0.x = fmaf(v2.x, f0.x, s0.x), s0.y = fmaf(v2.y, f0.y, s0.y);
      s1.x = fmaf(v2.x, f1.x, s1.x), s1.y = fmaf(v2.y, f1.y, s1.y);
      s2.x = fmaf(v2.x, f2.x, s2.x), s2.y = fmaf(v2.y, f2.y, s2.y);
    }
  };
  while (elo < emid && (int)spid[elo] < p_hi) edge(elo++);
  while (ehi < eend && (int)spid[ehi] < p_hi) edge(ehi++);
}
template <bool HAS_F>
__global__ void __launch_bounds__(FP_THREADS)
force_fwd_pairs_kernel(const float* __restrict__ phi1, const float* __restrict__ phi2, const float* __restrict__ geo,
                       const int* __restrict__ mol_ptr, const int* __restrict__ row_ptr, const int* __restrict__ col,
                       const int* __restrict__ pid, const float* __restrict__ f_in, float* __restrict__ f_out, int n_mol,
                       const int2* __restrict__ xg, const int* __restrict__ pair_ptr) {
  __shared__ __attribute__((aligned(16))) float fl[HAS_F ? NNHIP_MOL_STAGE_MAX * 3 * NF : 4];
  __shared__ __attribute__((aligned(16))) float buf[FP_PAIRS * (HAS_F ? 2 : 1) * NF];     // [which][pair][NF]
  __shared__ float4 sgeo[FP_MAX_EDGES];
  __shared__ unsigned short spid[FP_MAX_EDGES];
  __shared__ unsigned char scol[FP_MAX_EDGES];     // local sender, bit 7: a masked candidate (see force_fwd_kernel)
  const int b = xcd_tile(blockIdx.x, gridDim.x);
  if (b >= n_mol) return;
  const int a0 = mol_ptr[b], n = mol_ptr[b + 1] - a0;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  if (n > NNHIP_MOL_STAGE_MAX) {      // (uniform) a molecule this form cannot stage: a wave per row on global memory
    const int c4 = 4 * (lane & 31);
    const bool hi = lane >= 32;
    for (int i = a0 + wave; i < a0 + n; i += FP_THREADS / 64) {
      float4 acc[3];
#pragma unroll
      for (int k = 0; k < 3; ++k)
        acc[k] = (hi || !HAS_F) ? make_float4(0.f, 0.f, 0.f, 0.f) : ld4(f_in + ((size_t)i * 3 + k) * NF + c4);
      const int beg = row_ptr[i], end = row_ptr[i + 1];
      for (int e = beg; e < end; e += 2) {
        const int e1 = min(e + 1, end - 1);
        const float4 g0 = reinterpret_cast<const float4*>(geo)[e], g1 = reinterpret_cast<const float4*>(geo)[e1];
        const int p0 = pid[e], p1 = pid[e1];
        const int j0 = col[e], j1 = col[e1];
        const float4 g = hi ? g1 : g0;
        const size_t p = (size_t)(hi ? p1 : p0);
        const int j = hi ? j1 : j0;
        const int gz0 = xg ? xg[e].x : 0, gz1 = xg ? xg[e1].x : 0;
        if ((!hi || e + 1 < end) && (hi ? gz1 : gz0) != FT_ZERO_ROW) {
          const float4 v1 = ld4(phi1 + p * NF + c4);
          acc[0] = fma4(v1, g.x, acc[0]);
          acc[1] = fma4(v1, g.y, acc[1]);
          acc[2] = fma4(v1, g.z, acc[2]);
          if (HAS_F) {
            const float4 v2 = ld4(phi2 + p * NF + c4);
#pragma unroll
            for (int k = 0; k < 3; ++k) acc[k] = fma4(v2, ld4(f_in + ((size_t)j * 3 + k) * NF + c4), acc[k]);
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float4 o = add4(acc[k], upper_half(acc[k]));
        if (!hi) st4(f_out + ((size_t)i * 3 + k) * NF + c4, o);
      }
    }
    return;
  }
  const int E0 = row_ptr[a0], nE = row_ptr[a0 + n] - E0;
  const int P0 = pair_ptr[a0], nP = pair_ptr[a0 + n] - P0;
  // ---- stage the molecule's node rows and per-edge scalars
  if (HAS_F) {
    const float4* src = reinterpret_cast<const float4*>(f_in + (size_t)a0 * 3 * NF);
    for (int u = t; u < n * 3 * (NF / 4); u += FP_THREADS) reinterpret_cast<float4*>(fl)[u] = src[u];
  }
  for (int u = t; u < nE; u += FP_THREADS) {
    sgeo[u] = reinterpret_cast<const float4*>(geo)[E0 + u];
    spid[u] = (unsigned short)(pid[E0 + u] - P0);
    scol[u] = (unsigned char)((col[E0 + u] - a0) | ((xg && xg[E0 + u].x == FT_ZERO_ROW) ? 0x80 : 0));
  }
  // ---- this wave's rows (three: rows wave, wave + 8, wave + 16): sums in registers, two cursors each
  static_assert(FP_ROWS_PER_WAVE == 3, "three rows per wave are written out below");
  int lo0, mid0, hi0, end0, lo1, mid1, hi1, end1, lo2, mid2, hi2, end2;
  float2 a00, a01, a02, a10, a11, a12, a20, a21, a22;
  auto init_row = [&](const int k, int& lo, int& mid, int& hi_, int& end_, float2& s0, float2& s1, float2& s2) {
    const bool live = k < n;
    const int i = a0 + (live ? k : 0);
    const int beg = row_ptr[i], end = row_ptr[i + 1];
    lo = beg - E0;
    end_ = live ? end - E0 : lo;
    mid = hi_ = live ? end - (pair_ptr[i + 1] - pair_ptr[i]) - E0 : lo;
    const float2 z2 = make_float2(0.f, 0.f);
    s0 = (HAS_F && live) ? *reinterpret_cast<const float2*>(f_in + ((size_t)i * 3 + 0) * NF + 2 * lane) : z2;
    s1 = (HAS_F && live) ? *reinterpret_cast<const float2*>(f_in + ((size_t)i * 3 + 1) * NF + 2 * lane) : z2;
    s2 = (HAS_F && live) ? *reinterpret_cast<const float2*>(f_in + ((size_t)i * 3 + 2) * NF + 2 * lane) : z2;
  };
  init_row(wave, lo0, mid0, hi0, end0, a00, a01, a02);
  init_row(wave + FP_THREADS / 64, lo1, mid1, hi1, end1, a10, a11, a12);
  init_row(wave + 2 * (FP_THREADS / 64), lo2, mid2, hi2, end2, a20, a21, a22);
  // ---- the pair rows, FP_PAIRS at a time
  constexpr int UNITS = FP_PAIRS * (NF / 4) / FP_THREADS;       // float4 per thread, array and chunk
  float4 r1[UNITS], r2[UNITS];
#pragma unroll
  for (int q = 0; q < UNITS; ++q) r1[q] = r2[q] = make_float4(0.f, 0.f, 0.f, 0.f);
#define FP_REQUEST(c_)                                                                                \
  {                                                                                                   \
    const int cnt_ = min(FP_PAIRS, nP - (c_) * FP_PAIRS);                                             \
    const float4* s1_ = reinterpret_cast<const float4*>(phi1 + (size_t)(P0 + (c_) * FP_PAIRS) * NF);  \
    const float4* s2_ = reinterpret_cast<const float4*>(phi2 + (size_t)(P0 + (c_) * FP_PAIRS) * NF);  \
    _Pragma("unroll") for (int q = 0; q < UNITS; ++q) {                                               \
      const int u_ = t + q * FP_THREADS;                                                              \
      if (u_ < cnt_ * (NF / 4)) {                                                                     \
        r1[q] = s1_[u_];                                                                              \
        if (HAS_F) r2[q] = s2_[u_];                                                                   \
      }                                                                                               \
    }                                                                                                 \
  }
  const int n_chunks = (nP + FP_PAIRS - 1) / FP_PAIRS;
  if (n_chunks > 0) FP_REQUEST(0)
  for (int c = 0; c < n_chunks; ++c) {
    __syncthreads();                       // (the previous chunk is consumed; first trip: the staging above is complete)
#pragma unroll
    for (int q = 0; q < UNITS; ++q) {
      const int u = t + q * FP_THREADS;
      reinterpret_cast<float4*>(buf)[u] = r1[q];
      if (HAS_F) reinterpret_cast<float4*>(buf + FP_PAIRS * NF)[u] = r2[q];
    }
    __syncthreads();
    if (c + 1 < n_chunks) FP_REQUEST(c + 1)
    const int p_lo = c * FP_PAIRS, p_hi = p_lo + FP_PAIRS;
    fp_row_chunk<HAS_F>(lo0, mid0, hi0, end0, a00, a01, a02, p_lo, p_hi, lane, buf, fl, sgeo, spid, scol);
    fp_row_chunk<HAS_F>(lo1, mid1, hi1, end1, a10, a11, a12, p_lo, p_hi, lane, buf, fl, sgeo, spid, scol);
    fp_row_chunk<HAS_F>(lo2, mid2, hi2, end2, a20, a21, a22, p_lo, p_hi, lane, buf, fl, sgeo, spid, scol);
  }
  auto store_row = [&](const int k, const float2 s0, const float2 s1, const float2 s2) {
    if (k >= n) return;
    *reinterpret_cast<float2*>(f_out + ((size_t)(a0 + k) * 3 + 0) * NF + 2 * lane) = s0;
    *reinterpret_cast<float2*>(f_out + ((size_t)(a0 + k) * 3 + 1) * NF + 2 * lane) = s1;
    *reinterpret_cast<float2*>(f_out + ((size_t)(a0 + k) * 3 + 2) * NF + 2 * lane) = s2;
  };
  store_row(wave, a00, a01, a02);
  store_row(wave + FP_THREADS / 64, a10, a11, a12);
  store_row(wave + 2 * (FP_THREADS / 64), a20, a21, a22);
#undef FP_REQUEST
}

// ---------------------------------------------------------------------------------------------
// force_fwd for batches of small molecules with ALL rows of the molecule in flight at once: a HALF-wave per row (the lower half
// of a wave walks one row, the upper half another, an edge each per instruction), so the 24 rows of a molecule are one round of a
// 12-wave workgroup.  In force_fwd_mol_kernel a wave walks three rows one after the other and the two reads of a pair row -- by
// its two endpoints -- are a round (5-7 us, ~5 MB of the XCD's traffic) apart unless both rows are in the same round: 2/3 of the
// second reads miss the 4 MB L2 (counters: 322 MB fetched per launch against 193 needed).  Here both endpoints walk their
// neighbors in ascending order at the same pace and reach each other |i - j| * 2/3 trips apart.
// Per-edge scalars (geo, pair row, sender) come from an LDS copy: the two halves of a wave read different edges.
// ---------------------------------------------------------------------------------------------
#define FH_WAVES (NNHIP_MOL_STAGE_MAX / 2)
__global__ void __launch_bounds__(64 * FH_WAVES)
force_fwd_half_kernel(const float* __restrict__ phi1, const float* __restrict__ phi2, const float* __restrict__ geo,
                      const int* __restrict__ mol_ptr, const int* __restrict__ row_ptr, const int* __restrict__ col,
                      const int* __restrict__ pid, const float* __restrict__ f_in, float* __restrict__ f_out, int n_mol,
                      const int2* __restrict__ xg) {
  __shared__ __attribute__((aligned(16))) float fl[NNHIP_MOL_STAGE_MAX * 3 * NF];
  __shared__ float4 sgeo[FP_MAX_EDGES];
  __shared__ int spid[FP_MAX_EDGES];
  __shared__ unsigned char scol[FP_MAX_EDGES];     // local sender, bit 7: a masked candidate (see force_fwd_kernel)
  const int b = xcd_tile(blockIdx.x, gridDim.x);
  if (b >= n_mol) return;
  const int a0 = mol_ptr[b], n = mol_ptr[b + 1] - a0;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int c4 = 4 * (lane & 31);
  const bool hi = lane >= 32;
  if (n > NNHIP_MOL_STAGE_MAX) {      // (uniform) a molecule this form cannot stage: a wave per row on global memory
    for (int i = a0 + wave; i < a0 + n; i += FH_WAVES) {
      float4 acc[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) acc[k] = hi ? make_float4(0.f, 0.f, 0.f, 0.f) : ld4(f_in + ((size_t)i * 3 + k) * NF + c4);
      const int beg = row_ptr[i], end = row_ptr[i + 1];
      for (int e = beg; e < end; e += 2) {
        const int e1 = min(e + 1, end - 1);
        const float4 g0 = reinterpret_cast<const float4*>(geo)[e], g1 = reinterpret_cast<const float4*>(geo)[e1];
        const int p0 = pid[e], p1 = pid[e1];
        const int j0 = col[e], j1 = col[e1];
        const float4 g = hi ? g1 : g0;
        const size_t p = (size_t)(hi ? p1 : p0);
        const int j = hi ? j1 : j0;
        const int gz0 = xg ? xg[e].x : 0, gz1 = xg ? xg[e1].x : 0;
        if ((!hi || e + 1 < end) && (hi ? gz1 : gz0) != FT_ZERO_ROW) {
          const float4 v1 = ld4(phi1 + p * NF + c4), v2 = ld4(phi2 + p * NF + c4);
          acc[0] = fma4(v1, g.x, acc[0]);
          acc[1] = fma4(v1, g.y, acc[1]);
          acc[2] = fma4(v1, g.z, acc[2]);
#pragma unroll
          for (int k = 0; k < 3; ++k) acc[k] = fma4(v2, ld4(f_in + ((size_t)j * 3 + k) * NF + c4), acc[k]);
        }
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float4 o = add4(acc[k], upper_half(acc[k]));
        if (!hi) st4(f_out + ((size_t)i * 3 + k) * NF + c4, o);
      }
    }
    return;
  }
  const int E0 = row_ptr[a0], nE = row_ptr[a0 + n] - E0;
  {
    const float4* src = reinterpret_cast<const float4*>(f_in + (size_t)a0 * 3 * NF);
    for (int u = t; u < n * 3 * (NF / 4); u += 64 * FH_WAVES) reinterpret_cast<float4*>(fl)[u] = src[u];
  }
  for (int u = t; u < nE; u += 64 * FH_WAVES) {
    sgeo[u] = reinterpret_cast<const float4*>(geo)[E0 + u];
    spid[u] = pid[E0 + u];
    scol[u] = (unsigned char)((col[E0 + u] - a0) | ((xg && xg[E0 + u].x == FT_ZERO_ROW) ? 0x80 : 0));
  }
  const int k = 2 * wave + (hi ? 1 : 0);
  const bool live = k < n;
  const int beg = live ? row_ptr[a0 + k] - E0 : 0, end = live ? row_ptr[a0 + k + 1] - E0 : 0;
  const int len = end - beg;
  const int trips = max(len, __shfl_xor(len, 32, WAVE));     // (the longer of the wave's two rows)
  __syncthreads();
  float4 acc[3];
#pragma unroll
  for (int c = 0; c < 3; ++c)
    acc[c] = live ? *reinterpret_cast<const float4*>(fl + (k * 3 + c) * NF + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
  for (int q = 0; q < trips; ++q) {
    const int u = beg + q;
    if (u < end) {
      const int cj = scol[u];
      if (!(cj & 0x80)) {
        const size_t p = (size_t)spid[u];
        const float4 g = sgeo[u];
        const float4 v1 = ld4(phi1 + p * NF + c4), v2 = ld4(phi2 + p * NF + c4);
        acc[0] = fma4(v1, g.x, acc[0]);
        acc[1] = fma4(v1, g.y, acc[1]);
        acc[2] = fma4(v1, g.z, acc[2]);
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] = fma4(v2, *reinterpret_cast<const float4*>(fl + (cj * 3 + c) * NF + c4), acc[c]);
      }
    }
  }
  if (live) {
#pragma unroll
    for (int c = 0; c < 3; ++c) st4(f_out + ((size_t)(a0 + k) * 3 + c) * NF + c4, acc[c]);
  }
}


// ---- launcher lines (edge.hip:launch_force_fwd)
#if 0
  static const bool pairs_form = getenv("NNHIP_FORCE_FWD_MOL") && atoi(getenv("NNHIP_FORCE_FWD_MOL")) == 2;
  if (pairs_form && mol_ptr && pair_ptr && mol_kernels_pay(n_atoms, n_mol)) {
    if (has_f)
      force_fwd_pairs_kernel<true><<<n_mol, FP_THREADS, 0, s>>>(phi1, phi2, geo, mol_ptr, row_ptr, col, pid, f_in, f_out, n_mol,
                                                               reinterpret_cast<const int2*>(xg), pair_ptr);
    else
      force_fwd_pairs_kernel<false><<<n_mol, FP_THREADS, 0, s>>>(phi1, phi2, geo, mol_ptr, row_ptr, col, pid, f_in, f_out, n_mol,
                                                                reinterpret_cast<const int2*>(xg), pair_ptr);
    LAUNCH_CHECK();
    return 0;
  }
  static const bool half_form = getenv("NNHIP_FORCE_FWD_MOL") && atoi(getenv("NNHIP_FORCE_FWD_MOL")) == 3;
  if (half_form && has_f && mol_ptr && mol_kernels_pay(n_atoms, n_mol)) {
    force_fwd_half_kernel<<<n_mol, 64 * FH_WAVES, 0, s>>>(phi1, phi2, geo, mol_ptr, row_ptr, col, pid, f_in, f_out, n_mol,
                                                         reinterpret_cast<const int2*>(xg));
    LAUNCH_CHECK();
    return 0;
  }
#endif
