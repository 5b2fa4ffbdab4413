// EXPERIMENT (round 4, not part of the build): force_fwd with the NEXT trip's rows landing in LDS.
// Paste into csrc/edge.hip next to force_fwd_kernel and launch with EDGE_ROWS * 5 KiB of dynamic LDS
// (EDGE_ROWS * 1 KiB for HAS_F = false).  Measured: 0.191 -> 0.185 ms per step at config 2
// (profiles/r04_force_fwd_lds_prefetch_ab.txt); parity tests green.  DESIGN.md section 7 says what that means.
//
// The row kernels issue a trip's five row loads, wait a full L2 / fabric latency, compute, and only then issue the next trip's.
// Here the loads of trip t + 1 are issued BEFORE trip t is computed and land straight in a per-wave LDS slot
// (global_load_lds_dwordx4: lane-linear, 1 KiB per instruction; no register holds them while they travel), and the per-edge
// scalars (pid, col, geo, xg) of trip t + 2 are requested at the same point.
template <bool HAS_F, int WPR>
__global__ void __launch_bounds__(64 * EDGE_ROWS)
force_fwd_lds_kernel(const float* __restrict__ phi1, const float* __restrict__ phi2, const float* __restrict__ geo,
                     const int* __restrict__ row_ptr, const int* __restrict__ col, const int* __restrict__ pid,
                     const float* __restrict__ f_in, float* __restrict__ f_out, int n_atoms, const int2* __restrict__ xg) {
  constexpr int SLOTS = HAS_F ? 5 : 1;                 // phi1 | phi2 | f_in[j][0..2]: one 1 KiB row pair each
  extern __shared__ __attribute__((aligned(16))) float4 fwd_lds[];
  __shared__ float4 comb[EDGE_COMB_SIZE(WPR, 3)];
  int part;
  const int i_ = wave_row_split<WPR>(gridDim.x, part);
  const bool active = i_ < n_atoms;
  const int i = active ? i_ : 0;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c4 = 4 * (lane & 31);
  const bool hi = lane >= 32;
  float4* slot = fwd_lds + (size_t)wave * SLOTS * 64;
  float4 acc[3];
#pragma unroll
  for (int k = 0; k < 3; ++k)
    acc[k] = (HAS_F && !hi && part == 0) ? ld4(f_in + ((size_t)i * 3 + k) * NF + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
  const int beg = active ? row_ptr[i] : 0, end = active ? row_ptr[i + 1] : 0;
  struct Trip {
    int p0, p1, j0, j1, gz0, gz1;
    float4 g0, g1;
    bool two;
  };
  auto scalars = [&](int e) {
    Trip t;
    const int e1 = min(e + 1, end - 1);
    t.p0 = pid[e];
    t.p1 = pid[e1];
    t.j0 = HAS_F ? col[e] : 0;
    t.j1 = HAS_F ? col[e1] : 0;
    t.g0 = reinterpret_cast<const float4*>(geo)[e];
    t.g1 = reinterpret_cast<const float4*>(geo)[e1];
    t.gz0 = xg ? xg[e].x : 0;
    t.gz1 = xg ? xg[e1].x : 0;
    t.two = e + 1 < end;
    return t;
  };
  auto issue = [&](const Trip& t) {
    const size_t p = (size_t)(hi ? t.p1 : t.p0);
    __builtin_amdgcn_global_load_lds(phi1 + p * NF + c4, reinterpret_cast<float*>(slot), 16, 0, 0);
    if (HAS_F) {
      const size_t j = (size_t)(hi ? t.j1 : t.j0);
      __builtin_amdgcn_global_load_lds(phi2 + p * NF + c4, reinterpret_cast<float*>(slot + 64), 16, 0, 0);
#pragma unroll
      for (int k = 0; k < 3; ++k)
        __builtin_amdgcn_global_load_lds(f_in + (j * 3 + k) * NF + c4, reinterpret_cast<float*>(slot + (2 + k) * 64), 16, 0, 0);
    }
  };
  const int step = 2 * WPR;
  int e = beg + 2 * part;
  Trip cur, nxt;
  if (e < end) {
    cur = scalars(e);
    issue(cur);
    if (e + step < end) nxt = scalars(e + step);
  }
  for (; e < end; e += step) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this trip's rows have landed
    float4 v1 = slot[lane], v2, fj[3];
    if (HAS_F) {
      v2 = slot[64 + lane];
#pragma unroll
      for (int k = 0; k < 3; ++k) fj[k] = slot[(2 + k) * 64 + lane];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // ... and are in registers: the slot is free again
    const Trip now = cur;
    if (e + step < end) {
      issue(nxt);
      cur = nxt;
      if (e + 2 * step < end) nxt = scalars(e + 2 * step);
    }
    const float4 g = hi ? now.g1 : now.g0;
    const bool live = (!hi || now.two) && (hi ? now.gz1 : now.gz0) != FT_ZERO_ROW;
    if (live) {
      acc[0] = fma4(v1, g.x, acc[0]);
      acc[1] = fma4(v1, g.y, acc[1]);
      acc[2] = fma4(v1, g.z, acc[2]);
      if (HAS_F) {
#pragma unroll
        for (int k = 0; k < 3; ++k) acc[k] = fma4(v2, fj[k], acc[k]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) acc[k] = add4(acc[k], upper_half(acc[k]));
  row_combine<WPR, 3>(acc, comb, part, lane);
#pragma unroll
  for (int k = 0; k < 3; ++k)
    if (active && part == 0 && !hi) st4(f_out + ((size_t)i * 3 + k) * NF + c4, acc[k]);
}
