// Per-edge message-passing kernels and their adjoints (gfx950, fp32).
//
// Replaces the edge loop of InteractionNet.forward (newtonnet/models/newtonnet.py:207-231):
//   :210-215  msg = (W_e rbf_e) * m[i] * m[j];  atom_node += scatter_sum(msg, i)          -> msg_fwd_kernel
//   :219-227  force_node += scatter_sum(phi1_e (x) dir_e + phi2_e * force_node[j], i)     -> force_fwd_kernel
// and the reverse sweep torch.autograd.grad runs through them for the gradient force
// (newtonnet/models/output.py:66-73)                                                      -> *_bwd_kernel
//
// Layout / mapping: the edge list is a CSR over the receiver i (graph.hip).  A receiver row belongs to WPR = 1, 2 or 4
// wavefronts of one workgroup (edge_common.h, "Split rows": the waves of a row take alternate edge pairs of each of its two
// ranges and their partial row sums meet in LDS in wave order).  A wave walks its edges two at a time: the lower half-wave
// takes the even edge, the upper half the odd one, and lane l of a half holds features 4l .. 4l+3, so every [F]=128-float row access is a 16-byte-per-lane
// instruction covering two 512-byte rows (the texture addresser charges per instruction, not per byte).  The per-row
// sums live in registers (the two halves are folded once per row): deterministic segmented reductions, no float atomics.  Sender-side scatters of the adjoint are turned into receiver-side
// gathers (the edge set is symmetric).  Per-edge scalars (col, dir, table position) are wave-uniform and come
// in through the scalar cache.  HBM-bound: no MFMA here.
//
// Pair space: msg, the MLP hidden tiles and phi1 / phi2 are symmetric under i <-> j (graph.hip, "Undirected pairs"),
// so they are stored once per undirected pair p = pid[e] ([P = E/2][128] arrays).  The row of the LOWER endpoint owns
// the pair (its upper edges i < j map to a contiguous run of pair rows) and is the only writer of msg[p] and
// g_phi[p]; both endpoints read.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

#include "edge_common.h"

// ---------------------------------------------------------------------------------------------
// forward: message + invariant aggregation
//   msg[pid e] = eps_e * m[i] * m[j] (written by the row with i < j);  a_mid[i] = a_in[i] + sum_{e in row i} msg_e
// ---------------------------------------------------------------------------------------------
template <int WPR>
__global__ void __launch_bounds__(64 * EDGE_ROWS)
msg_fwd_kernel(const float* __restrict__ m, const int2* __restrict__ xg, const float* __restrict__ table,
               const int* __restrict__ row_ptr, const int* __restrict__ col, const int* __restrict__ pid,
               const float* __restrict__ a_in, float* __restrict__ msg /*[P][F]*/, float* __restrict__ a_mid,
               int n_atoms) {
  __shared__ float4 comb[EDGE_COMB_SIZE(WPR, 1)];
  int part;
  const int i_ = wave_row_split<WPR>(gridDim.x, part);
  const bool active = i_ < n_atoms;
  if (WPR == 1 && !active) return;
  const int i = active ? i_ : 0;            // (split rows: an idle wave still meets the others at the barrier of row_combine)
  const int lane = threadIdx.x & 63;
  const int c4 = 4 * (lane & 31);
  const bool hi = lane >= 32;
  const float4 mi = ld4(m + (size_t)i * NF + c4);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const int beg = active ? row_ptr[i] : 0, end = active ? row_ptr[i + 1] : 0;
  for (int e = beg + 2 * part; e < end; e += 2 * WPR) {
    const int e1 = min(e + 1, end - 1);
    const int j0 = col[e], j1 = col[e1];
    const int2 gx0 = xg[e], gx1 = xg[e1];   // wave-uniform
    const int jj = hi ? j1 : j0;
    const int2 gx = hi ? gx1 : gx0;
    if (!hi || e + 1 < end) {
      const FilterW fw = filter_weights(__int_as_float(gx.y));
#ifdef EDGE_ABL_HALF_TABLE   // tooling (wrong results): what a pair-centric evaluation of eps could save at most
      const float4 eps = jj > i ? filter_value(table, ABL_G(gx.x), c4, fw) : mi;
#elif defined(EDGE_ABL_NOTAB)   // tooling (wrong results): no table reads at all
      const float4 eps = mul4(mi, fw.w[0] + fw.w[1] + fw.w[2] + fw.w[3]);
#else
      const float4 eps = filter_value(table, ABL_G(gx.x), c4, fw);
#endif
#ifdef EDGE_ABL_NOMJ            // tooling (wrong results): no sender-row gather
      const float4 mj = mi;
#else
      const float4 mj = ld4(m + (size_t)ABL_J(jj, i) * NF + c4);
#endif
      const float4 v = mul4(mul4(eps, mi), mj);
      if (ABL_ST(jj > i)) {   // the lower endpoint writes the shared pair row
        const int p0 = pid[e], p1 = pid[e1];
        if (EDGE_NT_MSG)
          st4_nt(msg + (size_t)(hi ? p1 : p0) * NF + c4, v);
        else
          st4(msg + (size_t)(hi ? p1 : p0) * NF + c4, v);
      }
      acc = add4(acc, v);
    }
  }
  acc = add4(acc, upper_half(acc));
  {
    float4 a1[1] = {acc};
    row_combine<WPR, 1>(a1, comb, part, lane);
    acc = a1[0];
  }
  if (active && part == 0 && !hi) st4(a_mid + (size_t)i * NF + c4, add4(ld4(a_in + (size_t)i * NF + c4), acc));
}

// ---------------------------------------------------------------------------------------------
// forward: equivariant messages + aggregation
//   f_out[i][k] = f_in[i][k] + sum_e ( phi1[pid e] * u_e[k] + phi2[pid e] * f_in[j][k] )
// HAS_F = false for the first layer, where force_node == 0 (newtonnet.py:143): the phi2 term vanishes.
// ---------------------------------------------------------------------------------------------
template <bool HAS_F, int WPR>
__global__ void __launch_bounds__(64 * EDGE_ROWS)
force_fwd_kernel(const float* __restrict__ phi1 /*[P][F]*/, const float* __restrict__ phi2, const float* __restrict__ geo,
                 const int* __restrict__ row_ptr, const int* __restrict__ col, const int* __restrict__ pid,
                 const float* __restrict__ f_in, float* __restrict__ f_out, int n_atoms, const int2* __restrict__ xg,
                 const int* __restrict__ pair_ptr) {
  __shared__ float4 comb[EDGE_COMB_SIZE(WPR, 3)];
  int part;
  const int i_ = wave_row_split<WPR>(gridDim.x, part);
  const bool active = i_ < n_atoms;
  if (WPR == 1 && !active) return;
  const int i = active ? i_ : 0;
  const int lane = threadIdx.x & 63;
  const int c4 = 4 * (lane & 31);          // this lane's four features
  const bool hi = lane >= 32;              // upper half-wave: the odd edge of each pair of edges
  float4 acc[3];
#pragma unroll
  for (int k = 0; k < 3; ++k)
    acc[k] = (HAS_F && !hi && part == 0) ? ld4(f_in + ((size_t)i * 3 + k) * NF + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
  const int beg = active ? row_ptr[i] : 0, end = active ? row_ptr[i + 1] : 0;
  const int mid = row_mid_of(pair_ptr, col, beg, end, i, lane, active);
  auto run = [&](const int rb, const int re, auto nt) {   // nt: stream the pair rows (the other endpoint owns them)
    for (int e = rb + 2 * part; e < re; e += 2 * WPR) {
      const int e1 = min(e + 1, re - 1);    // (clamped; the odd half is masked off when the range has no edge e + 1)
      const float4 g0 = reinterpret_cast<const float4*>(geo)[e];   // (ux,uy,uz,r), wave-uniform
      const float4 g1 = reinterpret_cast<const float4*>(geo)[e1];
      const int p0 = pid[e], p1 = pid[e1];
      const float4 g = hi ? g1 : g0;
      const size_t p = (size_t)ABL_P(hi ? p1 : p0, i);
      // A candidate of a reused (Verlet-skin / static) list that is outside the cutoff right now (edge_embed_kernel points
      // its filter row at FT_ZERO_ROW, by the same fp32 predicate as the neighbor list) has msg == 0 but phi = W2 act(0),
      // which vanishes only when act(0) == 0 (not for sigmoid / softplus) -- mask it explicitly.  The caller passes
      // xg == NULL when act(0) == 0 (every contribution of such a candidate then vanishes by itself, and the per-edge
      // scalar loads of the mask cost 6 % of the edge time on a 54-neighbour periodic box)
      const int gz0 = xg ? xg[e].x : 0, gz1 = xg ? xg[e1].x : 0;   // (xg == NULL: nothing to mask)
      if ((!hi || e + 1 < re) && (hi ? gz1 : gz0) != FT_ZERO_ROW) {
        const float4 v1 = ld4p<decltype(nt)::value>(phi1 + p * NF + c4);
        acc[0] = fma4(v1, g.x, acc[0]);
        acc[1] = fma4(v1, g.y, acc[1]);
        acc[2] = fma4(v1, g.z, acc[2]);
        if (HAS_F) {
          const int j0 = col[e], j1 = col[e1];
          const int j = ABL_J(hi ? j1 : j0, i);
          const float4 v2 = ld4p<decltype(nt)::value>(phi2 + p * NF + c4);
#pragma unroll
          for (int k = 0; k < 3; ++k) acc[k] = fma4(v2, ld4(f_in + ((size_t)j * 3 + k) * NF + c4), acc[k]);
        }
      }
    }
  };
  run(beg, mid, std::integral_constant<bool, EDGE_NT_PHI_FWD != 0>());
  run(mid, end, std::false_type());
#pragma unroll
  for (int k = 0; k < 3; ++k) acc[k] = add4(acc[k], upper_half(acc[k]));
  row_combine<WPR, 3>(acc, comb, part, lane);
#pragma unroll
  for (int k = 0; k < 3; ++k)
    if (active && part == 0 && !hi) st4(f_out + ((size_t)i * 3 + k) * NF + c4, acc[k]);
}

// ---------------------------------------------------------------------------------------------
// force_fwd for batches of small molecules: one workgroup per MOLECULE with the molecule's f_in rows staged in LDS once.
// The sender rows f_in[j][0..2] are 1.5 of the 2.5 KB a directed edge pulls through the L2 -> L1 path, and a molecule's atoms
// gather the same rows ~14 times over; that stream, not HBM and not latency, bounds the row form of this kernel (DESIGN.md section 7:
// 76 -> 60 us per launch at config 2).  Same per-row arithmetic as force_fwd_kernel<true, 1>: a wave per row, two edges per
// instruction, the halves folded once.  A molecule of more than NNHIP_MOL_STAGE_MAX atoms (the caller launches this form only when
// the previous batch of the same shape had none: status bit 8 of the count pass) keeps gathering from global memory.
// ---------------------------------------------------------------------------------------------
#define FM_WAVES 8
// the next row of the molecule nobody has taken yet (one LDS atomic per row, by the wave's first lane)
__device__ __forceinline__ int fm_next_row(int* counter, int lane) {
  int k = 0;
  if (lane == 0) k = atomicAdd(counter, 1);
  return __builtin_amdgcn_readfirstlane(k);
}
__global__ void __launch_bounds__(64 * FM_WAVES)
force_fwd_mol_kernel(const float* __restrict__ phi1, const float* __restrict__ phi2, const float* __restrict__ geo,
                     const int* __restrict__ mol_ptr, const int* __restrict__ row_ptr, const int* __restrict__ col,
                     const int* __restrict__ pid, const float* __restrict__ f_in, float* __restrict__ f_out, int n_mol,
                     const int2* __restrict__ xg) {
  __shared__ __attribute__((aligned(16))) float fl[NNHIP_MOL_STAGE_MAX * 3 * NF];
  __shared__ int s_next;
  const int b = xcd_tile(blockIdx.x, gridDim.x);
  if (b >= n_mol) return;
  const int a0 = mol_ptr[b], n = mol_ptr[b + 1] - a0;
  const bool staged = n <= NNHIP_MOL_STAGE_MAX;
  if (staged) {
    const float4* src = reinterpret_cast<const float4*>(f_in + (size_t)a0 * 3 * NF);
    float4* dst = reinterpret_cast<float4*>(fl);
    for (int t = threadIdx.x; t < n * 3 * (NF / 4); t += 64 * FM_WAVES) dst[t] = src[t];
  }
  if (threadIdx.x == 0) s_next = FM_WAVES;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c4 = 4 * (lane & 31);
  const bool hi = lane >= 32;
  // rows are handed out as waves become free (the first FM_WAVES rows are taken; a row's sums do not depend on who forms them):
  // 21 rows of 10-20 edges on 8 waves in fixed turns leave the waves with two rows idle for a fifth of the workgroup's life
  for (int i = a0 + wave; i < a0 + n; i = a0 + fm_next_row(&s_next, lane)) {
    float4 acc[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
      acc[k] = hi ? make_float4(0.f, 0.f, 0.f, 0.f)
                  : (staged ? *reinterpret_cast<const float4*>(fl + ((i - a0) * 3 + k) * NF + c4) : ld4(f_in + ((size_t)i * 3 + k) * NF + c4));
    const int beg = row_ptr[i], end = row_ptr[i + 1];
    for (int e = beg; e < end; e += 2) {
      const int e1 = min(e + 1, end - 1);
      const float4 g0 = reinterpret_cast<const float4*>(geo)[e];
      const float4 g1 = reinterpret_cast<const float4*>(geo)[e1];
      const int p0 = pid[e], p1 = pid[e1];
      const int j0 = col[e], j1 = col[e1];
      const float4 g = hi ? g1 : g0;
      const size_t p = (size_t)(hi ? p1 : p0);
      const int j = hi ? j1 : j0;
      const int gz0 = xg ? xg[e].x : 0, gz1 = xg ? xg[e1].x : 0;   // (xg == NULL: nothing to mask; see force_fwd_kernel)
      if ((!hi || e + 1 < end) && (hi ? gz1 : gz0) != FT_ZERO_ROW) {
        // (the pair rows the other endpoint owns with the streaming hint, as the row form: step 1.491 -> 1.481 ms same box; the hint on
        // EVERY pair row: this kernel 0.160 -> 0.182 ms per step -- profiles/r04_force_fwd_mol_nt_ab.txt)
        const bool nt_ = j < i;
        const float4 v1 = nt_ ? ld4_nt(phi1 + p * NF + c4) : ld4(phi1 + p * NF + c4);
        const float4 v2 = nt_ ? ld4_nt(phi2 + p * NF + c4) : ld4(phi2 + p * NF + c4);
        acc[0] = fma4(v1, g.x, acc[0]);
        acc[1] = fma4(v1, g.y, acc[1]);
        acc[2] = fma4(v1, g.z, acc[2]);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float4 fj = staged ? *reinterpret_cast<const float4*>(fl + ((j - a0) * 3 + k) * NF + c4)
                                   : ld4(f_in + ((size_t)j * 3 + k) * NF + c4);
          acc[k] = fma4(v2, fj, acc[k]);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float4 o = add4(acc[k], upper_half(acc[k]));
      if (!hi) st4(f_out + ((size_t)i * 3 + k) * NF + c4, o);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// adjoint of force_fwd for receiver row i, given gf = dE/d f_out (p = pid[e], the shared pair row):
//   g_u[e][k]   = < gf[i][k] , phi1[p] >                                  (wave reduction, per directed edge)
//   g_fin[i][k] = gf[i][k] + sum_{e in row i} phi2[p] * gf[j][k]
//                 (the sender-side scatter  g_fin[j] += phi2[e] * gf[i]  seen from the receiving end of the reverse
//                  edge, whose phi2 is the same pair row)
//   and, written once per pair by the row of the lower endpoint (i < j), the sum of both directions' contributions
//   to the shared phi rows -- u_(j,i) = -u_(i,j):
//   g_phi1[p]   = sum_k (gf[i][k] - gf[j][k]) u_e[k]                      -> g_h12[p][0:F]   (feeds the MLP adjoint)
//   g_phi2[p]   = sum_k gf[i][k] * f_in[j][k] + gf[j][k] * f_in[i][k]     -> g_h12[p][F:2F]
// ---------------------------------------------------------------------------------------------
// (second launch bound: at least 4 waves per SIMD, i.e. at most 128 registers -- the split-row forms otherwise come out at 130 and
// lose a quarter of their occupancy)
template <bool HAS_F, int WPR>
__global__ void __launch_bounds__(64 * EDGE_ROWS, 4)
force_bwd_kernel(const float* __restrict__ gf, const float* __restrict__ phi1, const float* __restrict__ phi2,
                 const float* __restrict__ geo, const int* __restrict__ row_ptr, const int* __restrict__ col,
                 const int* __restrict__ pid, const float* __restrict__ f_in, float* __restrict__ g_h12 /*[P][2F]*/,
                 float* __restrict__ g_u /*[E][4]: gux,guy,guz,(unused)*/, float* __restrict__ g_fin, int n_atoms,
                 const int2* __restrict__ xg, const int* __restrict__ pair_ptr, const int* __restrict__ rev) {
  // rev != NULL (round 6; every caller that has the reverse-edge index): the pair's OWNER writes g_u of BOTH directed edges --
  // g_u[rev e][k] = < gf[j][k], phi1[p] > from the gf[j] rows it gathers anyway -- so the other endpoint never reads phi1[p]:
  // one pair-row read less per pair (80 of 560 MB per launch at config 2), and layer 0 (no phi2) visits only the pairs a row owns.
  // Same operands, same lane layout, same reduction: the bits of g_u do not change.
  __shared__ float4 comb[EDGE_COMB_SIZE(WPR, 3)];
  int part;
  const int i_ = wave_row_split<WPR>(gridDim.x, part);
  const bool active = i_ < n_atoms;
  if (WPR == 1 && !active) return;
  const int i = active ? i_ : 0;
  const int lane = threadIdx.x & 63;
  const int c4 = 4 * (lane & 31);
  const bool hi = lane >= 32;
  float4 gfi[3], fi[3], acc[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    gfi[k] = ld4(gf + ((size_t)i * 3 + k) * NF + c4);
    acc[k] = (hi || part != 0) ? make_float4(0.f, 0.f, 0.f, 0.f) : gfi[k];
    if (HAS_F) fi[k] = ld4(f_in + ((size_t)i * 3 + k) * NF + c4);
  }
  const int beg = active ? row_ptr[i] : 0, end = active ? row_ptr[i + 1] : 0;
  const int mid = row_mid_of(pair_ptr, col, beg, end, i, lane, active);
  const bool owner_gu = rev != nullptr;   // (uniform)
  // [beg, mid): pairs owned by the other endpoint -- the phi2 gather (and, without rev, this direction's g_u)
  for (int e = beg + 2 * part; e < mid && (HAS_F || !owner_gu); e += 2 * WPR) {
    const int e1 = min(e + 1, mid - 1);
    const int p0 = pid[e], p1 = pid[e1];
    const int eh = hi ? e1 : e;
    const size_t p = (size_t)(hi ? p1 : p0);
    const int gz0 = xg ? xg[e].x : 0, gz1 = xg ? xg[e1].x : 0;   // (xg == NULL: nothing to mask)
    const bool inside = (hi ? gz1 : gz0) != FT_ZERO_ROW;   // (see force_fwd_kernel: candidates outside the cutoff contribute nothing)
    if (!owner_gu && (!hi || e + 1 < mid) && !inside && (lane & 31) == 31)
      reinterpret_cast<float4*>(g_u)[eh] = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((!hi || e + 1 < mid) && inside) {
      float4 gfj[3];
      if (HAS_F) {
        const int j0 = col[e], j1 = col[e1];
        const int j = ABL_J(hi ? j1 : j0, i);
        const float4 v2 = ld4p<EDGE_NT_PHI_BWD != 0>(phi2 + (size_t)ABL_P(p, i) * NF + c4);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          gfj[k] = ld4(gf + ((size_t)j * 3 + k) * NF + c4);
          acc[k] = fma4(v2, gfj[k], acc[k]);
        }
      }
      if (!owner_gu) {
        const float4 v1 = ld4p<EDGE_NT_PHI_BWD != 0>(phi1 + (size_t)ABL_P(p, i) * NF + c4);
        const float s0 = half_sum_top(dot4(gfi[0], v1));
        const float s1 = half_sum_top(dot4(gfi[1], v1));
        const float s2 = half_sum_top(dot4(gfi[2], v1));
        if ((lane & 31) == 31) reinterpret_cast<float4*>(g_u)[eh] = make_float4(s0, s1, s2, 0.f);
      }
    }
  }
  // [mid, end): pairs this row owns -- additionally the adjoints of the shared phi rows
  for (int e = mid + 2 * part; e < end; e += 2 * WPR) {
    const int e1 = min(e + 1, end - 1);
    const int p0 = pid[e], p1 = pid[e1];
    const int j0 = col[e], j1 = col[e1];
    const float4 g0 = reinterpret_cast<const float4*>(geo)[e];
    const float4 g1 = reinterpret_cast<const float4*>(geo)[e1];
    const int eh = hi ? e1 : e;
    const size_t p = (size_t)(hi ? p1 : p0);
    const int j = ABL_J(hi ? j1 : j0, i);
    const float4 g = hi ? g1 : g0;
    const int gz0 = xg ? xg[e].x : 0, gz1 = xg ? xg[e1].x : 0;   // (xg == NULL: nothing to mask)
    const bool inside = (hi ? gz1 : gz0) != FT_ZERO_ROW;
    const int er0 = owner_gu ? rev[e] : 0, er1 = owner_gu ? rev[e1] : 0;   // (the reverse edges: wave-uniform loads)
    const int er = hi ? er1 : er0;
    if ((!hi || e + 1 < end) && !inside) {   // outside the cutoff: zero adjoints for the pair rows this row owns
      const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
      st4(g_h12 + p * 2 * NF + c4, zero);
      if (HAS_F) st4(g_h12 + p * 2 * NF + NF + c4, zero);
      if ((lane & 31) == 31) {
        reinterpret_cast<float4*>(g_u)[eh] = zero;
        if (owner_gu) reinterpret_cast<float4*>(g_u)[er] = zero;
      }
    }
    if ((!hi || e + 1 < end) && inside) {
      const float4 v1 = ld4(phi1 + (size_t)ABL_P(p, i) * NF + c4);
      float4 gfj[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) gfj[k] = ld4(gf + ((size_t)j * 3 + k) * NF + c4);
      float4 gp1 = mul4(sub4(gfi[0], gfj[0]), g.x);
      gp1 = fma4(sub4(gfi[1], gfj[1]), g.y, gp1);
      gp1 = fma4(sub4(gfi[2], gfj[2]), g.z, gp1);
      if (ABL_ST(true)) { if (EDGE_NT_GH) st4_nt(g_h12 + p * 2 * NF + c4, gp1); else st4(g_h12 + p * 2 * NF + c4, gp1); }
      if (HAS_F) {
        const float4 v2 = ld4(phi2 + (size_t)ABL_P(p, i) * NF + c4);
        float4 gp2 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          acc[k] = fma4(v2, gfj[k], acc[k]);
          gp2 = fma4(gfi[k], ld4(f_in + ((size_t)j * 3 + k) * NF + c4), gp2);
          gp2 = fma4(gfj[k], fi[k], gp2);
        }
        if (ABL_ST(true)) { if (EDGE_NT_GH) st4_nt(g_h12 + p * 2 * NF + NF + c4, gp2); else st4(g_h12 + p * 2 * NF + NF + c4, gp2); }
      }
      const float s0 = half_sum_top(dot4(gfi[0], v1));
      const float s1 = half_sum_top(dot4(gfi[1], v1));
      const float s2 = half_sum_top(dot4(gfi[2], v1));
      if ((lane & 31) == 31) reinterpret_cast<float4*>(g_u)[eh] = make_float4(s0, s1, s2, 0.f);
      if (owner_gu) {      // the other direction's g_u: what row j computed from gf[j] and this same phi1 row
        const float t0 = half_sum_top(dot4(gfj[0], v1));
        const float t1 = half_sum_top(dot4(gfj[1], v1));
        const float t2 = half_sum_top(dot4(gfj[2], v1));
        if ((lane & 31) == 31) reinterpret_cast<float4*>(g_u)[er] = make_float4(t0, t1, t2, 0.f);
      }
    }
  }
  if (HAS_F) {
#pragma unroll
    for (int k = 0; k < 3; ++k) acc[k] = add4(acc[k], upper_half(acc[k]));
    row_combine<WPR, 3>(acc, comb, part, lane);
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (active && part == 0 && !hi) st4(g_fin + ((size_t)i * 3 + k) * NF + c4, acc[k]);
  }
}

// ---------------------------------------------------------------------------------------------
// adjoint of msg_fwd for receiver row i.  g_msg[p] holds the MLP-side gradient of the shared pair message; the
// aggregations a_mid[i] += msg and a_mid[j] += msg (one per directed edge) add g_a[i] + g_a[j]:
//   G       = g_msg[p] + g_a[i] + g_a[j]                 total gradient of the pair message
//   g_m[i]  = sum_{e in row i} G * eps_e * m[j]
//   g_x[e]  = < G * m[i] * m[j] , d eps_e/dx >  for j > i, 0 for j < i   (wave reduction; x is shared by the two
//             directed edges and only their sum enters the force, so the pair's owner carries all of it and the
//             other direction skips the derivative table)
// ---------------------------------------------------------------------------------------------
// NEED_GM = false for the first layer: its m = message_nodepart(Embedding[z]) does not depend on the positions, so g_m is
// never used and the rows only visit the pairs they own (for g_x).
template <bool NEED_GM, int WPR>
__global__ void __launch_bounds__(64 * EDGE_ROWS)
msg_bwd_kernel(const float* __restrict__ g_msg /*[P][F]*/, const float* __restrict__ g_a, const float* __restrict__ m,
               const int2* __restrict__ xg, const float* __restrict__ table,
               const int* __restrict__ row_ptr, const int* __restrict__ col, const int* __restrict__ pid,
               float* __restrict__ g_m, float* __restrict__ g_x, int n_atoms, const int* __restrict__ pair_ptr) {
  __shared__ float4 comb[EDGE_COMB_SIZE(WPR, 1)];
  int part;
  const int i_ = wave_row_split<WPR>(gridDim.x, part);
  const bool active = i_ < n_atoms;
  if (WPR == 1 && !active) return;
  const int i = active ? i_ : 0;
  const int lane = threadIdx.x & 63;
  const int c4 = 4 * (lane & 31);
  const bool hi = lane >= 32;
  const float4 mi = ld4(m + (size_t)i * NF + c4);
  const float4 gai = ld4(g_a + (size_t)i * NF + c4);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const int beg = active ? row_ptr[i] : 0, end = active ? row_ptr[i + 1] : 0;
  const int mid = row_mid_of(pair_ptr, col, beg, end, i, lane, active);
  if (part == 0)
    for (int e = beg + lane; e < mid; e += 64) g_x[e] = 0.f;   // the pair's owner carries all of g_x
  if (NEED_GM) {
    // [beg, mid): pairs owned by the other endpoint -- value table only
    for (int e = beg + 2 * part; e < mid; e += 2 * WPR) {
      const int e1 = min(e + 1, mid - 1);
      const int j0 = col[e], j1 = col[e1];
      const int p0 = pid[e], p1 = pid[e1];
      const int2 gx0 = xg[e], gx1 = xg[e1];
      const int j = ABL_J(hi ? j1 : j0, i);
      const size_t p = (size_t)ABL_P(hi ? p1 : p0, i);
      const int2 gx = hi ? gx1 : gx0;
      if (!hi || e + 1 < mid) {
        const float4 mj = ld4(m + (size_t)j * NF + c4);
        const float4 gaj = ld4(g_a + (size_t)j * NF + c4);
        const float4 G = add4(add4(ld4p<EDGE_NT_GMSG != 0>(g_msg + p * NF + c4), gai), gaj);
        const FilterW fw = filter_weights(__int_as_float(gx.y));
#ifdef EDGE_ABL_HALF_TABLE
        const float4 eps = mi;
#else
        const float4 eps = filter_value(table, ABL_G(gx.x), c4, fw);
#endif
        acc = fma4(mul4(G, eps), mj, acc);
      }
    }
  }
  // [mid, end): pairs this row owns -- value and derivative, g_x
  for (int e = mid + 2 * part; e < end; e += 2 * WPR) {
    const int e1 = min(e + 1, end - 1);
    const int j0 = col[e], j1 = col[e1];
    const int p0 = pid[e], p1 = pid[e1];
    const int2 gx0 = xg[e], gx1 = xg[e1];
    const int j = ABL_J(hi ? j1 : j0, i);
    const size_t p = (size_t)ABL_P(hi ? p1 : p0, i);
    const int2 gx = hi ? gx1 : gx0;
    const int eh = hi ? e1 : e;
    if (!hi || e + 1 < end) {
      const float4 mj = ld4(m + (size_t)j * NF + c4);
      const float4 gaj = ld4(g_a + (size_t)j * NF + c4);
      const float4 G = add4(add4(ld4(g_msg + p * NF + c4), gai), gaj);
      const FilterW fw = filter_weights(__int_as_float(gx.y));
      float4 eps, deps;
      filter_value_deriv(table, ABL_G(gx.x), c4, fw, eps, deps);
      const float gxs = half_sum_top(dot4(mul4(mul4(G, mi), mj), deps));
      if ((lane & 31) == 31) g_x[eh] = gxs;
      if (NEED_GM) acc = fma4(mul4(G, eps), mj, acc);
    }
  }
  if (NEED_GM) {
    acc = add4(acc, upper_half(acc));
    float4 a1[1] = {acc};
    row_combine<WPR, 1>(a1, comb, part, lane);
    if (active && part == 0 && !hi) st4(g_m + (size_t)i * NF + c4, a1[0]);
  }
}

// msg_bwd for batches of small molecules: one workgroup per molecule, the molecule's m and g_a rows staged in LDS once (the
// same idea as force_fwd_mol_kernel: the sender-row gathers m[j], g_a[j] are 1 KB of the 3.5 KB a directed edge pulls through the
// vector memory path).  Per-row arithmetic of msg_bwd_kernel<NEED_GM, 1>.
template <bool NEED_GM>
__global__ void __launch_bounds__(64 * FM_WAVES)
msg_bwd_mol_kernel(const float* __restrict__ g_msg /*[P][F]*/, const float* __restrict__ g_a, const float* __restrict__ m,
                   const int2* __restrict__ xg, const float* __restrict__ table, const int* __restrict__ mol_ptr,
                   const int* __restrict__ row_ptr, const int* __restrict__ col, const int* __restrict__ pid,
                   float* __restrict__ g_m, float* __restrict__ g_x, int n_mol, const int* __restrict__ pair_ptr) {
  __shared__ __attribute__((aligned(16))) float ml[NNHIP_MOL_STAGE_MAX * NF];
  __shared__ __attribute__((aligned(16))) float gl[NNHIP_MOL_STAGE_MAX * NF];
  __shared__ int s_next;
  const int b = xcd_tile(blockIdx.x, gridDim.x);
  if (b >= n_mol) return;
  const int a0 = mol_ptr[b], n = mol_ptr[b + 1] - a0;
  const bool staged = n <= NNHIP_MOL_STAGE_MAX;
  if (staged) {
    const float4* sm = reinterpret_cast<const float4*>(m + (size_t)a0 * NF);
    const float4* sg = reinterpret_cast<const float4*>(g_a + (size_t)a0 * NF);
    for (int t = threadIdx.x; t < n * (NF / 4); t += 64 * FM_WAVES) {
      reinterpret_cast<float4*>(ml)[t] = sm[t];
      reinterpret_cast<float4*>(gl)[t] = sg[t];
    }
  }
  if (threadIdx.x == 0) s_next = FM_WAVES;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c4 = 4 * (lane & 31);
  const bool hi = lane >= 32;
  auto node = [&](const float* lds, const float* glob, int j) {
    return staged ? *reinterpret_cast<const float4*>(lds + (j - a0) * NF + c4) : ld4(glob + (size_t)j * NF + c4);
  };
  for (int i = a0 + wave; i < a0 + n; i = a0 + fm_next_row(&s_next, lane)) {   // (rows handed out as waves become free)
    const float4 mi = node(ml, m, i), gai = node(gl, g_a, i);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int beg = row_ptr[i], end = row_ptr[i + 1];
    const int mid = row_mid_of(pair_ptr, col, beg, end, i, lane, true);
    for (int e = beg + lane; e < mid; e += 64) g_x[e] = 0.f;   // the pair's owner carries all of g_x
    if (NEED_GM) {
      for (int e = beg; e < mid; e += 2) {     // pairs owned by the other endpoint -- value table only
        const int e1 = min(e + 1, mid - 1);
        const int j0 = col[e], j1 = col[e1];
        const int p0 = pid[e], p1 = pid[e1];
        const int2 gx0 = xg[e], gx1 = xg[e1];
        const int j = hi ? j1 : j0;
        const size_t p = (size_t)(hi ? p1 : p0);
        const int2 gx = hi ? gx1 : gx0;
        if (!hi || e + 1 < mid) {
          const float4 mj = node(ml, m, j), gaj = node(gl, g_a, j);
          const float4 G = add4(add4(ld4p<EDGE_NT_GMSG != 0>(g_msg + p * NF + c4), gai), gaj);
          const FilterW fw = filter_weights(__int_as_float(gx.y));
          const float4 eps = filter_value(table, gx.x, c4, fw);
          acc = fma4(mul4(G, eps), mj, acc);
        }
      }
    }
    for (int e = mid; e < end; e += 2) {       // pairs this row owns -- value and derivative, g_x
      const int e1 = min(e + 1, end - 1);
      const int j0 = col[e], j1 = col[e1];
      const int p0 = pid[e], p1 = pid[e1];
      const int2 gx0 = xg[e], gx1 = xg[e1];
      const int j = hi ? j1 : j0;
      const size_t p = (size_t)(hi ? p1 : p0);
      const int2 gx = hi ? gx1 : gx0;
      const int eh = hi ? e1 : e;
      if (!hi || e + 1 < end) {
        const float4 mj = node(ml, m, j), gaj = node(gl, g_a, j);
        const float4 G = add4(add4(ld4(g_msg + p * NF + c4), gai), gaj);
        const FilterW fw = filter_weights(__int_as_float(gx.y));
        float4 eps, deps;
        filter_value_deriv(table, gx.x, c4, fw, eps, deps);
        const float gxs = half_sum_top(dot4(mul4(mul4(G, mi), mj), deps));
        if ((lane & 31) == 31) g_x[eh] = gxs;
        if (NEED_GM) acc = fma4(mul4(G, eps), mj, acc);
      }
    }
    if (NEED_GM) {
      acc = add4(acc, upper_half(acc));
      if (!hi) st4(g_m + (size_t)i * NF + c4, acc);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// geometry adjoint -> forces (and virial).  For edge e = (i,j) with disp d = pos_i - pos_j, r = |d|, u = d/r,
// x = r/rc:   g_d[e] = (g_x/rc) u + (g_u - (g_u.u) u)/r,  summed over layers.  pos_i enters row i's edges
// with +1 and the reverse edges with -1:  dE/dpos_i = sum_{e in row i} (g_d[e] - g_d[rev e]);  force = -that.
// The strain derivative (virial, output.py:154-165) is  -sum_e d_e (x) g_d[e]  per molecule.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
edge_gd_kernel(const float* __restrict__ g_x /*[L][E]*/, const float* __restrict__ g_u /*[L][E][4]*/,
               const float* __restrict__ geo, int n_edges, int n_layers, float inv_rc, float* __restrict__ g_d /*[E][4]*/) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_edges) return;
  float gx = 0.f, gu0 = 0.f, gu1 = 0.f, gu2 = 0.f;
  for (int l = 0; l < n_layers; ++l) {
    gx += g_x[(size_t)l * n_edges + e];
    const float4 v = reinterpret_cast<const float4*>(g_u)[(size_t)l * n_edges + e];
    gu0 += v.x;
    gu1 += v.y;
    gu2 += v.z;
  }
  const float4 g = reinterpret_cast<const float4*>(geo)[e];
  const float ir = 1.0f / g.w;
  const float dot = gu0 * g.x + gu1 * g.y + gu2 * g.z;
  const float a = gx * inv_rc - dot * ir;
  reinterpret_cast<float4*>(g_d)[e] =
      make_float4(fmaf(a, g.x, gu0 * ir), fmaf(a, g.y, gu1 * ir), fmaf(a, g.z, gu2 * ir), 0.f);
}

__global__ void __launch_bounds__(256)
force_out_kernel(const float* __restrict__ g_d, const int* __restrict__ row_ptr, const int* __restrict__ rev,
                 int n_atoms, float* __restrict__ forces) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_atoms) return;
  float fx = 0.f, fy = 0.f, fz = 0.f;
  for (int e = row_ptr[i]; e < row_ptr[i + 1]; ++e) {
    const float4 a = reinterpret_cast<const float4*>(g_d)[e];
    const float4 b = reinterpret_cast<const float4*>(g_d)[rev[e]];
    fx -= (a.x - b.x);
    fy -= (a.y - b.y);
    fz -= (a.z - b.z);
  }
  forces[3 * (size_t)i] = fx;
  forces[3 * (size_t)i + 1] = fy;
  forces[3 * (size_t)i + 2] = fz;
}

// edge_gd + force_out in one launch when nobody needs g_d itself (no virial): 16 lanes per atom walk the row, each evaluating
// g_d of its edge AND of the reverse edge from the per-layer g_x / g_u; the 16 partial forces are folded in a fixed order.
__device__ __forceinline__ float4 gd_of_edge(const float* __restrict__ g_x, const float* __restrict__ g_u,
                                             const float* __restrict__ geo, int e, int n_edges, int n_layers, float inv_rc) {
  float gx = 0.f, gu0 = 0.f, gu1 = 0.f, gu2 = 0.f;
  for (int l = 0; l < n_layers; ++l) {
    gx += g_x[(size_t)l * n_edges + e];
    const float4 v = reinterpret_cast<const float4*>(g_u)[(size_t)l * n_edges + e];
    gu0 += v.x;
    gu1 += v.y;
    gu2 += v.z;
  }
  const float4 g = reinterpret_cast<const float4*>(geo)[e];
  const float ir = 1.0f / g.w;
  const float dot = gu0 * g.x + gu1 * g.y + gu2 * g.z;
  const float a = gx * inv_rc - dot * ir;
  return make_float4(fmaf(a, g.x, gu0 * ir), fmaf(a, g.y, gu1 * ir), fmaf(a, g.z, gu2 * ir), 0.f);
}
__global__ void __launch_bounds__(256)
force_direct_kernel(const float* __restrict__ g_x, const float* __restrict__ g_u, const float* __restrict__ geo,
                    const int* __restrict__ row_ptr, const int* __restrict__ rev, int n_atoms, int n_edges, int n_layers,
                    float inv_rc, float* __restrict__ forces) {
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int sub = threadIdx.x & 15;
  float fx = 0.f, fy = 0.f, fz = 0.f;
  if (i < n_atoms) {
    for (int e = row_ptr[i] + sub; e < row_ptr[i + 1]; e += 16) {
      const float4 a = gd_of_edge(g_x, g_u, geo, e, n_edges, n_layers, inv_rc);
      const float4 b = gd_of_edge(g_x, g_u, geo, rev[e], n_edges, n_layers, inv_rc);
      fx -= (a.x - b.x);
      fy -= (a.y - b.y);
      fz -= (a.z - b.z);
    }
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {   // fixed butterfly inside the 16-lane group: deterministic
    fx += __shfl_xor(fx, o, WAVE);
    fy += __shfl_xor(fy, o, WAVE);
    fz += __shfl_xor(fz, o, WAVE);
  }
  if (i < n_atoms && sub == 0) {
    forces[3 * (size_t)i] = fx;
    forces[3 * (size_t)i + 1] = fy;
    forces[3 * (size_t)i + 2] = fz;
  }
}

// virial[b] = -dE/d(displacement_b)  (output.py:154-165), the derivative w.r.t. the symmetric strain S the reference
// applies as pos @ S and cell @ S (newtonnet.py:153-155).  With d_e = (pos_i - pos_j) S - (cell S) n_e, exactly as
// RadiusGraph writes the image shift (representations.py:93):
//     dE/dS[a][b] = sum_e ( dp_e[a] g_d[e][b] - (cell^T g_d[e])[a] n_e[b] ),   virial = -sym(dE/dS)
// n_e (the integer image) is recovered from dp_e - disp_e = cell n_e.  One wave per molecule, lane per atom row.
__global__ void __launch_bounds__(64)
virial_kernel(const float* __restrict__ g_d, const float* __restrict__ disp, const float* __restrict__ pos,
              const float* __restrict__ cell, const int* __restrict__ row_ptr, const int* __restrict__ col,
              const int* __restrict__ mol_ptr, int n_mol, float* __restrict__ virial) {
  const int b = blockIdx.x;
  if (b >= n_mol) return;
  const int lane = threadIdx.x;
  double c[9], ic[9];
  bool pbc = false;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    c[k] = (double)cell[(size_t)b * 9 + k];
    pbc |= (c[k] != 0.0);
  }
  if (pbc) {
    const double c00 = c[4] * c[8] - c[5] * c[7], c01 = c[5] * c[6] - c[3] * c[8], c02 = c[3] * c[7] - c[4] * c[6];
    const double id = 1.0 / (c[0] * c00 + c[1] * c01 + c[2] * c02);
    ic[0] = c00 * id; ic[1] = (c[2] * c[7] - c[1] * c[8]) * id; ic[2] = (c[1] * c[5] - c[2] * c[4]) * id;
    ic[3] = c01 * id; ic[4] = (c[0] * c[8] - c[2] * c[6]) * id; ic[5] = (c[2] * c[3] - c[0] * c[5]) * id;
    ic[6] = c02 * id; ic[7] = (c[1] * c[6] - c[0] * c[7]) * id; ic[8] = (c[0] * c[4] - c[1] * c[3]) * id;
  }
  double s[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) s[k] = 0.0;
  for (int i = mol_ptr[b] + lane; i < mol_ptr[b + 1]; i += 64) {
    const double pi[3] = {pos[3 * (size_t)i], pos[3 * (size_t)i + 1], pos[3 * (size_t)i + 2]};
    for (int e = row_ptr[i]; e < row_ptr[i + 1]; ++e) {
      const int j = col[e];
      const float4 g4 = reinterpret_cast<const float4*>(g_d)[e];
      const double g[3] = {g4.x, g4.y, g4.z};
      double dp[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) dp[a] = pi[a] - (double)pos[3 * (size_t)j + a];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int q = 0; q < 3; ++q) s[a * 3 + q] += dp[a] * g[q];
      if (pbc) {
        double sh[3], n[3], ctg[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) sh[a] = dp[a] - (double)disp[3 * (size_t)e + a];   // = (cell n)[a]
#pragma unroll
        for (int a = 0; a < 3; ++a) n[a] = rint(ic[a * 3] * sh[0] + ic[a * 3 + 1] * sh[1] + ic[a * 3 + 2] * sh[2]);
#pragma unroll
        for (int a = 0; a < 3; ++a) ctg[a] = c[a] * g[0] + c[3 + a] * g[1] + c[6 + a] * g[2];  // (cell^T g)[a]
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int q = 0; q < 3; ++q) s[a * 3 + q] -= ctg[a] * n[q];
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    double v = s[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    s[k] = v;
  }
  if (lane == 0) {
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int q = 0; q < 3; ++q) virial[(size_t)b * 9 + p * 3 + q] = (float)(-0.5 * (s[p * 3 + q] + s[q * 3 + p]));
  }
}

// ---------------------------------------------------------------------------------------------
// node-level elementwise pieces (N x F; an order of magnitude less traffic than the edge tensors)
// ---------------------------------------------------------------------------------------------
// atom_node = Embedding[z]   (newtonnet.py:142)
// and, in the same pass, the first layer's m = message_nodepart(Embedding[z]) looked up from its per-element table
// (pipeline.hip evaluates that MLP on the 119 embedding rows instead of the N atom rows)
__global__ void embed_kernel(const int64_t* __restrict__ z, const float* __restrict__ table,
                             const float* __restrict__ m_table, int n_atoms, float* __restrict__ a0,
                             float* __restrict__ m0) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // one float4 each
  if (t >= (size_t)n_atoms * (NF / 4)) return;
  const int i = (int)(t / (NF / 4)), c = (int)(t % (NF / 4));
  const size_t zi = (size_t)clamp_species(z[i]);
  reinterpret_cast<float4*>(a0)[t] = reinterpret_cast<const float4*>(table + zi * NF)[c];
  reinterpret_cast<float4*>(m0)[t] = reinterpret_cast<const float4*>(m_table + zi * NF)[c];
}

// LayerNorm over the 128 features of atom_node at the end of an interaction layer (nn.LayerNorm(n_features), eps 1e-5,
// newtonnet.py:202-205,228-231), in place, keeping x_hat and 1/sigma for the adjoint:
//   x_hat = (x - mean) / sqrt(var + eps);  y = x_hat * gamma + beta
//   dE/dx = (1/sigma) (g' - mean(g') - x_hat mean(g' x_hat)),   g' = dE/dy * gamma
__global__ void __launch_bounds__(256)
layer_norm_fwd_kernel(float* __restrict__ a /*in: pre-norm, out: normalised*/, const float* __restrict__ gamma,
                      const float* __restrict__ beta, int n_atoms, float* __restrict__ xhat, float* __restrict__ rstd) {
  const int i = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (i >= n_atoms) return;
  const int lane = threadIdx.x & 63;
  const float2 x = ld2(a + (size_t)i * NF + 2 * lane);
  const float mean = wave_sum(x.x + x.y) * (1.0f / NF);
  const float dx = x.x - mean, dy = x.y - mean;
  const float var = wave_sum(fmaf(dx, dx, dy * dy)) * (1.0f / NF);
  const float rs = 1.0f / sqrtf(var + 1e-5f);
  const float2 xh = make_float2(dx * rs, dy * rs);
  const float2 g = ld2(gamma + 2 * lane), b = ld2(beta + 2 * lane);
  st2(xhat + (size_t)i * NF + 2 * lane, xh);
  st2(a + (size_t)i * NF + 2 * lane, make_float2(fmaf(xh.x, g.x, b.x), fmaf(xh.y, g.y, b.y)));
  if (lane == 0) rstd[i] = rs;
}
__global__ void __launch_bounds__(256)
layer_norm_bwd_kernel(float* __restrict__ g_a /*in: dE/dy, out: dE/dx*/, const float* __restrict__ gamma,
                      const float* __restrict__ xhat, const float* __restrict__ rstd, int n_atoms) {
  const int i = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (i >= n_atoms) return;
  const int lane = threadIdx.x & 63;
  const float2 g = ld2(g_a + (size_t)i * NF + 2 * lane), w = ld2(gamma + 2 * lane);
  const float2 xh = ld2(xhat + (size_t)i * NF + 2 * lane);
  const float2 gp = make_float2(g.x * w.x, g.y * w.y);
  const float m1 = wave_sum(gp.x + gp.y) * (1.0f / NF);
  const float m2 = wave_sum(fmaf(gp.x, xh.x, gp.y * xh.y)) * (1.0f / NF);
  const float rs = rstd[i];
  st2(g_a + (size_t)i * NF + 2 * lane, make_float2(rs * (gp.x - m1 - xh.x * m2), rs * (gp.y - m1 - xh.y * m2)));
}

// energy head tail (output.py:98-100 last Linear, scalers.py:55-58) and the seed of the reverse sweep:
//   eps_i = <silu(e2_i), w4> + b4;  E_i = eps_i * scale[z_i] + shift[z_i]
//   g_e2[i] = scale[z_i] * w4 * silu'(e2_i)          (dE_b/dE_i = 1, output.py:69)
__global__ void __launch_bounds__(256)
head_out_kernel(const float* __restrict__ e2, const float* __restrict__ w4, const float* __restrict__ b4,
                const float* __restrict__ scale, const float* __restrict__ shift, const int64_t* __restrict__ z,
                int n_atoms, int act, float* __restrict__ atom_energy, float* __restrict__ g_e2) {
  const int i = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (i >= n_atoms) return;
  const int lane = threadIdx.x & 63;
  const float2 h = ld2(e2 + (size_t)i * NF + 2 * lane);
  const float2 w = ld2(w4 + 2 * lane);
  const bool silu = act == NNHIP_ACT_SILU;   // uniform
  const float ax = silu ? silu_f(h.x) : act_f(h.x, act), ay = silu ? silu_f(h.y) : act_f(h.y, act);
  const float s = wave_sum(fmaf(ax, w.x, ay * w.y));
  const long zi = clamp_species(z[i]);
  const float sc = scale ? scale[zi] : 1.0f;
  const float sh = shift ? shift[zi] : 0.0f;
  if (lane == 0) atom_energy[i] = fmaf(s + b4[0], sc, sh);
  if (g_e2) {
    const float dx = silu ? dsilu_f(h.x) : dact_f(h.x, act), dy = silu ? dsilu_f(h.y) : dact_f(h.y, act);
    st2(g_e2 + (size_t)i * NF + 2 * lane, make_float2(sc * w.x * dx, sc * w.y * dy));
  }
}

// E_b = sum of atom energies of molecule b  (output.py:246).  One wave per molecule, fp64 partial sums in a fixed
// lane-strided order + butterfly: deterministic, one rounding at the end; a 100k-atom box no longer serialises on one
// thread.
// A molecule of many thousand atoms (the 100k-atom box is ONE molecule) would serialise on one wave (380 us there): molecules
// longer than MOL_SPLIT atoms are summed by mol_energy_big_kernel, one 1024-thread workgroup per molecule, fp64 partial sums in a
// fixed thread-strided order + a fixed tree: deterministic, one rounding at the end.
#define MOL_SPLIT 4096
__global__ void __launch_bounds__(1024)
mol_energy_big_kernel(const float* __restrict__ atom_energy, const int* __restrict__ mol_ptr, int n_mol, float* __restrict__ energy) {
  __shared__ double sh[1024];
  const int b = blockIdx.x;
  const int beg = mol_ptr[b], end = mol_ptr[b + 1];
  if (end - beg <= MOL_SPLIT) return;     // (short molecules: mol_energy_kernel)
  double s = 0.0;
  for (int i = beg + threadIdx.x; i < end; i += 1024) s += (double)atom_energy[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) energy[b] = (float)sh[0];
}
__global__ void __launch_bounds__(256)
mol_energy_kernel(const float* __restrict__ atom_energy, const int* __restrict__ mol_ptr, int n_mol,
                  float* __restrict__ energy, int long_from) {
  const int b = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (b >= n_mol) return;
  const int lane = threadIdx.x & 63;
  if (mol_ptr[b + 1] - mol_ptr[b] > long_from) return;   // (left to mol_energy_big_kernel)
  double s = 0.0;
  for (int i = mol_ptr[b] + lane; i < mol_ptr[b + 1]; i += 64) s += (double)atom_energy[i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, WAVE);
  if (lane == 0) energy[b] = (float)s;
}

// head_out_kernel + mol_energy_kernel in one launch for batches of small molecules: a workgroup per molecule, a wave per atom
// (the same arithmetic per atom), then the molecule's first wave sums the atom energies its workgroup has just written -- the
// same lane-strided fp64 partial sums and butterfly as mol_energy_kernel: bit for bit the same energies.
#define HM_WAVES 8
__global__ void __launch_bounds__(64 * HM_WAVES)
head_out_mol_kernel(const float* __restrict__ e2, const float* __restrict__ w4, const float* __restrict__ b4,
                    const float* __restrict__ scale, const float* __restrict__ shift, const int64_t* __restrict__ z,
                    const int* __restrict__ mol_ptr, int act, float* atom_energy, float* __restrict__ g_e2,
                    float* __restrict__ energy) {
  const int b = blockIdx.x;
  const int beg = mol_ptr[b], end = mol_ptr[b + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float2 w = ld2(w4 + 2 * lane);
  const bool silu = act == NNHIP_ACT_SILU;   // uniform
  for (int i = beg + wave; i < end; i += HM_WAVES) {
    const float2 h = ld2(e2 + (size_t)i * NF + 2 * lane);
    const float ax = silu ? silu_f(h.x) : act_f(h.x, act), ay = silu ? silu_f(h.y) : act_f(h.y, act);
    const float s = wave_sum(fmaf(ax, w.x, ay * w.y));
    const long zi = clamp_species(z[i]);
    const float sc = scale ? scale[zi] : 1.0f;
    const float sh = shift ? shift[zi] : 0.0f;
    if (lane == 0) atom_energy[i] = fmaf(s + b4[0], sc, sh);
    if (g_e2) {
      const float dx = silu ? dsilu_f(h.x) : dact_f(h.x, act), dy = silu ? dsilu_f(h.y) : dact_f(h.y, act);
      st2(g_e2 + (size_t)i * NF + 2 * lane, make_float2(sc * w.x * dx, sc * w.y * dy));
    }
  }
  __syncthreads();     // (the atom energies of this molecule are written: same workgroup, lines nobody has read before)
  if (wave == 0) {
    double s = 0.0;
    for (int i = beg + lane; i < end; i += 64) s += (double)atom_energy[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, WAVE);
    if (lane == 0) energy[b] = (float)s;
  }
}

// out[m][n][k] = in[m][k][n] for a list of 128x128 matrices (weights for the adjoint GEMMs)
struct TransposeList {
  const float* src[40];
  float* dst[40];
};
__global__ void __launch_bounds__(256) transpose128_kernel(TransposeList L) {
  __shared__ float tile[32][33];
  const float* __restrict__ src = L.src[blockIdx.z];
  float* __restrict__ dst = L.dst[blockIdx.z];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) tile[r][tx] = src[(size_t)(by + r) * NF + bx + tx];
  __syncthreads();
  for (int r = ty; r < 32; r += 8) dst[(size_t)(bx + r) * NF + by + tx] = tile[tx][r];
}

// ---------------------------------------------------------------------------------------------
// host-side launchers (used by pipeline.hip)
// ---------------------------------------------------------------------------------------------
static inline int row_blocks(int n_atoms, int wpr) { return cdiv(n_atoms, EDGE_ROWS / wpr); }
// Small systems (the one-molecule MD step, small training batches) are bound by the latency chain of a row, not by traffic or
// occupancy: up to EDGE_SMALL_ATOMS rows every kernel gives a row four waves (NNHIP_EDGE_SMALL_ATOMS overrides, 0 = never).
// Back-to-back aspirin batches, us per step with / without: 1008 atoms 389 / 404, 2016: 303 / 304, 3024: 390 / 383, 5376: 574 / 568
// (profiles/r04_small_thresholds.txt).
#ifndef EDGE_SMALL_ATOMS
#define EDGE_SMALL_ATOMS 2048
#endif
static inline bool edge_small(int n_atoms) {
  static const int lim = getenv("NNHIP_EDGE_SMALL_ATOMS") ? atoi(getenv("NNHIP_EDGE_SMALL_ATOMS")) : EDGE_SMALL_ATOMS;
  return n_atoms <= lim;
}
// NNHIP_EDGE_WPR=1|2|4 (read once per process) forces one split for every row kernel at every size: the non-default forms stay
// reachable for tests (tests/test_hip_parity.py::test_non_default_forms_in_child_processes) without a tooling build.
static inline int edge_wpr_forced() {
  static const int v = [] {
    const char* e = getenv("NNHIP_EDGE_WPR");
    const int w = e ? atoi(e) : 0;
    return (w == 1 || w == 2 || w == 4) ? w : 0;
  }();
  return v;
}
#define EDGE_LAUNCH_W(KERNEL_W, W_, ...) KERNEL_W<<<row_blocks(n_atoms, W_), 64 * EDGE_ROWS, edge_lds(), s>>>(__VA_ARGS__)
#define EDGE_LAUNCH(KERNEL, WPR_, ...)                                                                          \
  do {                                                                                                          \
    const int w_ = edge_wpr_forced() ? edge_wpr_forced() : (edge_small(n_atoms) ? 4 : WPR_);                    \
    if (w_ == 4)                                                                                                \
      EDGE_LAUNCH_W(KERNEL<4>, 4, __VA_ARGS__);                                                                 \
    else if (w_ == 2)                                                                                           \
      EDGE_LAUNCH_W(KERNEL<2>, 2, __VA_ARGS__);                                                                 \
    else                                                                                                        \
      EDGE_LAUNCH_W(KERNEL<1>, 1, __VA_ARGS__);                                                                 \
  } while (0)
#define EDGE_LAUNCH_B(KERNEL, FLAG, WPR_, ...)                                                                  \
  do {                                                                                                          \
    const int w_ = edge_wpr_forced() ? edge_wpr_forced() : (edge_small(n_atoms) ? 4 : WPR_);                    \
    if (w_ == 4)                                                                                                \
      EDGE_LAUNCH_W((KERNEL<FLAG, 4>), 4, __VA_ARGS__);                                                         \
    else if (w_ == 2)                                                                                           \
      EDGE_LAUNCH_W((KERNEL<FLAG, 2>), 2, __VA_ARGS__);                                                         \
    else                                                                                                        \
      EDGE_LAUNCH_W((KERNEL<FLAG, 1>), 1, __VA_ARGS__);                                                         \
  } while (0)
// tooling: NNHIP_EDGE_LDS=<bytes> attaches unused dynamic LDS to the edge kernels to cap their occupancy
static inline size_t edge_lds() {
  static const size_t v = getenv("NNHIP_EDGE_LDS") ? (size_t)atol(getenv("NNHIP_EDGE_LDS")) : 0;
  return v;
}

// The molecule-resident kernels (force_fwd_mol_kernel, msg_bwd_mol_kernel: one 8-wave workgroup per molecule) need enough molecules
// to fill the chip: same box, us per step with / without them -- 100 molecules 370 / 314, 256: 561 / 529, 384: 675 / 654,
// 512: 845 / 842, 768: 1151 / 1169, 1024: 1535 / 1585 (profiles/r04_mol_kernels_crossover.txt).  NNHIP_MOL_KERNELS_MIN overrides.
// ... and molecules large enough to have rows for the eight waves: 21.5k atoms in molecules of 3 / 6 / 9 / 12 / 16 / 21 atoms,
// us per step with / without -- 694 / 672, 885 / 879, 1102 / 1107, 1307 / 1311, 1492 / 1518, 1781 / 1834
// (profiles/r04_mol_kernels_by_molecule_size.txt): from an average of 8 atoms per molecule up.
static bool mol_kernels_pay(int n_atoms, int n_mol) {
  static const int min_mol = getenv("NNHIP_MOL_KERNELS_MIN") ? atoi(getenv("NNHIP_MOL_KERNELS_MIN")) : 640;
  return n_mol >= min_mol && (long)n_atoms <= (long)n_mol * NNHIP_MOL_STAGE_MAX && (long)n_atoms >= 8L * n_mol;
}

// what the launchers below decide with, for nnhip_config (pipeline.hip)
void edge_config(int* small_atoms, int* mol_min, int* wpr /*[4]: msg_fwd, force_fwd, force_bwd, msg_bwd*/, int* mol_forms /*bits: force_fwd, msg_bwd, force_direct, head_out*/) {
  *small_atoms = getenv("NNHIP_EDGE_SMALL_ATOMS") ? atoi(getenv("NNHIP_EDGE_SMALL_ATOMS")) : EDGE_SMALL_ATOMS;
  *mol_min = getenv("NNHIP_MOL_KERNELS_MIN") ? atoi(getenv("NNHIP_MOL_KERNELS_MIN")) : 640;
  wpr[0] = EDGE_WPR_MSG_FWD, wpr[1] = EDGE_WPR_FORCE_FWD, wpr[2] = EDGE_WPR_FORCE_BWD, wpr[3] = EDGE_WPR_MSG_BWD;
  auto off = [](const char* name) { return getenv(name) && atoi(getenv(name)) == 0; };
  *mol_forms = (off("NNHIP_FORCE_FWD_MOL") ? 0 : 1) | (off("NNHIP_MSG_BWD_MOL") ? 0 : 2) | (off("NNHIP_FORCE_DIRECT_MOL") ? 0 : 4) |
               (off("NNHIP_HEAD_OUT_MOL") ? 0 : 8);
}

int launch_msg_fwd(const float* m, const int* xg, const float* table, const int* row_ptr, const int* col,
                   const int* pid, const float* a_in, float* msg, float* a_mid, int n_atoms, hipStream_t s) {
  ScopedTimer t0(TC_EDGE, s);
  ScopedTimer t1(TC_EDGE_FWD_MSG, s);
  EDGE_LAUNCH(msg_fwd_kernel, EDGE_WPR_MSG_FWD, m, reinterpret_cast<const int2*>(xg), table, row_ptr, col, pid, a_in, msg, a_mid, n_atoms);
  LAUNCH_CHECK();
  return 0;
}

int launch_force_fwd(bool has_f, const float* phi1, const float* phi2, const float* geo, const int* row_ptr,
                     const int* col, const int* pid, const float* f_in, float* f_out, int n_atoms, const int* xg,
                     hipStream_t s, const int* pair_ptr, const int* mol_ptr, int n_mol) {
  ScopedTimer t0(TC_EDGE, s);
  ScopedTimer t1(TC_EDGE_FWD_FORCE, s);
  // batches of small molecules (the caller passes mol_ptr only when it may: see force_fwd_mol_kernel); NNHIP_FORCE_FWD_MOL=0: never
  static const bool mol_off = getenv("NNHIP_FORCE_FWD_MOL") && atoi(getenv("NNHIP_FORCE_FWD_MOL")) == 0;
  if (has_f && mol_ptr && mol_kernels_pay(n_atoms, n_mol) && !mol_off) {
    // (eight waves: 4 workgroups x 36 KB of LDS = the CU's 32 wave slots; 7 waves -- no idle slot in the last round of a 21-atom
    // molecule -- 0.198 against 0.159 ms per step, 16 waves 0.173: profiles/r04_force_fwd_mol_waves_ab.txt)
    force_fwd_mol_kernel<<<n_mol, 64 * FM_WAVES, 0, s>>>(phi1, phi2, geo, mol_ptr, row_ptr, col, pid, f_in, f_out, n_mol,
                                                        reinterpret_cast<const int2*>(xg));
    LAUNCH_CHECK();
    return 0;
  }
  if (has_f)
    EDGE_LAUNCH_B(force_fwd_kernel, true, EDGE_WPR_FORCE_FWD, phi1, phi2, geo, row_ptr, col, pid, f_in, f_out, n_atoms, reinterpret_cast<const int2*>(xg), pair_ptr);
  else
    EDGE_LAUNCH_B(force_fwd_kernel, false, EDGE_WPR_FORCE_FWD, phi1, phi2, geo, row_ptr, col, pid, f_in, f_out, n_atoms, reinterpret_cast<const int2*>(xg), pair_ptr);
  LAUNCH_CHECK();
  return 0;
}

int launch_force_bwd(bool has_f, const float* gf, const float* phi1, const float* phi2, const float* geo,
                     const int* row_ptr, const int* col, const int* pid, const float* f_in, float* g_h12, float* g_u,
                     float* g_fin, int n_atoms, const int* xg, hipStream_t s, const int* pair_ptr, const int* rev) {
  ScopedTimer t0(TC_EDGE, s);
  ScopedTimer t1(TC_EDGE_BWD_FORCE, s);
  static const bool owner_off = getenv("NNHIP_FORCE_BWD_OWNER_GU") && atoi(getenv("NNHIP_FORCE_BWD_OWNER_GU")) == 0;   // (A/B)
  if (owner_off) rev = nullptr;
  if (has_f)
    EDGE_LAUNCH_B(force_bwd_kernel, true, EDGE_WPR_FORCE_BWD, gf, phi1, phi2, geo, row_ptr, col, pid, f_in, g_h12, g_u, g_fin, n_atoms, reinterpret_cast<const int2*>(xg), pair_ptr, rev);
  else
    EDGE_LAUNCH_B(force_bwd_kernel, false, EDGE_WPR_FORCE_BWD, gf, phi1, phi2, geo, row_ptr, col, pid, f_in, g_h12, g_u, g_fin, n_atoms, reinterpret_cast<const int2*>(xg), pair_ptr, rev);
  LAUNCH_CHECK();
  return 0;
}

int launch_msg_bwd(const float* g_msg, const float* g_a, const float* m, const int* xg, const float* table,
                   const int* row_ptr, const int* col, const int* pid, float* g_m, float* g_x, int n_atoms,
                   bool need_gm, hipStream_t s, const int* pair_ptr, const int* mol_ptr, int n_mol) {
  ScopedTimer t0(TC_EDGE, s);
  ScopedTimer t1(TC_EDGE_BWD_MSG, s);
  // batches of small molecules (see launch_force_fwd); NNHIP_MSG_BWD_MOL=0: never
  static const bool mol_off = getenv("NNHIP_MSG_BWD_MOL") && atoi(getenv("NNHIP_MSG_BWD_MOL")) == 0;
  if (mol_ptr && pair_ptr && mol_kernels_pay(n_atoms, n_mol) && !mol_off) {
    if (need_gm)
      msg_bwd_mol_kernel<true><<<n_mol, 64 * FM_WAVES, 0, s>>>(g_msg, g_a, m, reinterpret_cast<const int2*>(xg), table, mol_ptr, row_ptr,
                                                              col, pid, g_m, g_x, n_mol, pair_ptr);
    else
      msg_bwd_mol_kernel<false><<<n_mol, 64 * FM_WAVES, 0, s>>>(g_msg, g_a, m, reinterpret_cast<const int2*>(xg), table, mol_ptr, row_ptr,
                                                               col, pid, g_m, g_x, n_mol, pair_ptr);
    LAUNCH_CHECK();
    return 0;
  }
  if (need_gm)
    EDGE_LAUNCH_B(msg_bwd_kernel, true, EDGE_WPR_MSG_BWD, g_msg, g_a, m, reinterpret_cast<const int2*>(xg), table, row_ptr, col, pid, g_m, g_x, n_atoms, pair_ptr);
  else
    EDGE_LAUNCH_B(msg_bwd_kernel, false, EDGE_WPR_MSG_BWD, g_msg, g_a, m, reinterpret_cast<const int2*>(xg), table, row_ptr, col, pid, g_m, g_x, n_atoms, pair_ptr);
  LAUNCH_CHECK();
  return 0;
}

// force_direct_kernel for batches of small molecules: a workgroup per molecule forms g_d of each of the molecule's edges ONCE
// into LDS (the row form evaluates it from both ends: every edge's per-layer g_x / g_u is read twice, the second time through
// rev[]), then the same 16 lanes per atom add the same terms in the same order -- bit for bit the row form's forces.
#define FD_MAX_EDGES (NNHIP_MOL_STAGE_MAX * (NNHIP_MOL_STAGE_MAX - 1))
__global__ void __launch_bounds__(256)
force_direct_mol_kernel(const float* __restrict__ g_x, const float* __restrict__ g_u, const float* __restrict__ geo,
                        const int* __restrict__ mol_ptr, const int* __restrict__ row_ptr, const int* __restrict__ rev, int n_edges,
                        int n_layers, float inv_rc, float* __restrict__ forces) {
  __shared__ float4 sgd[FD_MAX_EDGES];
  const int b = blockIdx.x;
  const int a0 = mol_ptr[b], n = mol_ptr[b + 1] - a0;
  const bool staged = n <= NNHIP_MOL_STAGE_MAX;
  const int E0 = row_ptr[a0];
  if (staged) {
    const int nE = row_ptr[a0 + n] - E0;
    for (int u = threadIdx.x; u < nE; u += 256) sgd[u] = gd_of_edge(g_x, g_u, geo, E0 + u, n_edges, n_layers, inv_rc);
  }
  __syncthreads();
  const int sub = threadIdx.x & 15;
  for (int k0 = 0; k0 < n; k0 += 16) {
    const int k = k0 + (threadIdx.x >> 4);
    const int i = a0 + k;
    float fx = 0.f, fy = 0.f, fz = 0.f;
    if (k < n) {
      for (int e = row_ptr[i] + sub; e < row_ptr[i + 1]; e += 16) {
        const float4 a = staged ? sgd[e - E0] : gd_of_edge(g_x, g_u, geo, e, n_edges, n_layers, inv_rc);
        const float4 c = staged ? sgd[rev[e] - E0] : gd_of_edge(g_x, g_u, geo, rev[e], n_edges, n_layers, inv_rc);
        fx -= (a.x - c.x);
        fy -= (a.y - c.y);
        fz -= (a.z - c.z);
      }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      fx += __shfl_xor(fx, o, WAVE);
      fy += __shfl_xor(fy, o, WAVE);
      fz += __shfl_xor(fz, o, WAVE);
    }
    if (k < n && sub == 0) {
      forces[3 * (size_t)i] = fx;
      forces[3 * (size_t)i + 1] = fy;
      forces[3 * (size_t)i + 2] = fz;
    }
  }
}

int launch_geometry_bwd(const float* g_x, const float* g_u, const float* geo, const float* disp, const float* pos,
                        const float* cell, const int* row_ptr, const int* col, const int* rev, const int* mol_ptr,
                        int n_atoms, int n_edges, int n_mol, int n_layers, float cutoff, float* g_d, float* forces,
                        float* virial, hipStream_t s, bool small_molecules) {
  ScopedTimer t0(TC_OTHER, s);
  static const bool mol_off = getenv("NNHIP_FORCE_DIRECT_MOL") && atoi(getenv("NNHIP_FORCE_DIRECT_MOL")) == 0;
  if (!virial && small_molecules && mol_ptr && n_mol > 0 && (long)n_atoms <= (long)n_mol * NNHIP_MOL_STAGE_MAX && !mol_off) {
    force_direct_mol_kernel<<<n_mol, 256, 0, s>>>(g_x, g_u, geo, mol_ptr, row_ptr, rev, n_edges, n_layers, 1.0f / cutoff, forces);
    LAUNCH_CHECK();
    return 0;
  }
  // (the one-launch form evaluates g_d twice per directed edge, once from each side: cheaper than a launch for molecular batches,
  // 1.6x dearer than the two launches on the 100k-atom box with its 54 neighbours per atom -- measured 441 vs ~280 us)
  if (!virial && n_edges <= (1 << 20)) {   // nobody reads g_d: one launch
    force_direct_kernel<<<cdiv((long)n_atoms * 16, 256), 256, 0, s>>>(g_x, g_u, geo, row_ptr, rev, n_atoms, n_edges, n_layers,
                                                                     1.0f / cutoff, forces);
    LAUNCH_CHECK();
    return 0;
  }
  if (n_edges > 0) {
    edge_gd_kernel<<<cdiv(n_edges, 256), 256, 0, s>>>(g_x, g_u, geo, n_edges, n_layers, 1.0f / cutoff, g_d);
    LAUNCH_CHECK();
  }
  force_out_kernel<<<cdiv(n_atoms, 64), 64, 0, s>>>(g_d, row_ptr, rev, n_atoms, forces);
  LAUNCH_CHECK();
  if (virial) {
    virial_kernel<<<n_mol, 64, 0, s>>>(g_d, disp, pos, cell, row_ptr, col, mol_ptr, n_mol, virial);
    LAUNCH_CHECK();
  }
  return 0;
}

int launch_layer_norm_fwd(float* a, const float* gamma, const float* beta, int n_atoms, float* xhat, float* rstd,
                          hipStream_t s) {
  ScopedTimer t0(TC_OTHER, s);
  layer_norm_fwd_kernel<<<cdiv(n_atoms, ROWS_PER_BLOCK), 256, 0, s>>>(a, gamma, beta, n_atoms, xhat, rstd);
  LAUNCH_CHECK();
  return 0;
}
int launch_layer_norm_bwd(float* g_a, const float* gamma, const float* xhat, const float* rstd, int n_atoms,
                          hipStream_t s) {
  ScopedTimer t0(TC_OTHER, s);
  layer_norm_bwd_kernel<<<cdiv(n_atoms, ROWS_PER_BLOCK), 256, 0, s>>>(g_a, gamma, xhat, rstd, n_atoms);
  LAUNCH_CHECK();
  return 0;
}

int launch_embed(const int64_t* z, const float* table, const float* m_table, int n_atoms, float* a0, float* m0,
                 hipStream_t s) {
  ScopedTimer t0(TC_OTHER, s);
  embed_kernel<<<cdiv((long)n_atoms * (NF / 4), 256), 256, 0, s>>>(z, table, m_table, n_atoms, a0, m0);
  LAUNCH_CHECK();
  return 0;
}

int launch_head_out(const float* e2, const float* w4, const float* b4, const float* scale, const float* shift,
                    const int64_t* z, const int* mol_ptr, int n_atoms, int n_mol, int act, float* atom_energy, float* g_e2,
                    float* energy, hipStream_t s, bool small_molecules) {
  ScopedTimer t0(TC_OTHER, s);
  static const bool mol_off = getenv("NNHIP_HEAD_OUT_MOL") && atoi(getenv("NNHIP_HEAD_OUT_MOL")) == 0;
  if (small_molecules && n_mol > 0 && (long)n_atoms <= (long)n_mol * NNHIP_MOL_STAGE_MAX && !mol_off) {
    head_out_mol_kernel<<<n_mol, 64 * HM_WAVES, 0, s>>>(e2, w4, b4, scale, shift, z, mol_ptr, act, atom_energy, g_e2, energy);
    LAUNCH_CHECK();
    return 0;
  }
  head_out_kernel<<<cdiv(n_atoms, ROWS_PER_BLOCK), 256, 0, s>>>(e2, w4, b4, scale, shift, z, n_atoms, act, atom_energy, g_e2);
  LAUNCH_CHECK();
  // some molecule MAY be long (the host knows only the totals, unless the count pass said so: small_molecules); few molecules: cheap
  const bool big = n_atoms > MOL_SPLIT && n_mol <= 4096 && !small_molecules;
  mol_energy_kernel<<<cdiv(n_mol, ROWS_PER_BLOCK), 256, 0, s>>>(atom_energy, mol_ptr, n_mol, energy, big ? MOL_SPLIT : 0x7fffffff);
  LAUNCH_CHECK();
  if (big) {
    mol_energy_big_kernel<<<n_mol, 1024, 0, s>>>(atom_energy, mol_ptr, n_mol, energy);
    LAUNCH_CHECK();
  }
  return 0;
}

int launch_transposes(const float* const* src, float* const* dst, int count, hipStream_t s) {
  ScopedTimer t0(TC_OTHER, s);
  TransposeList L;
  if (count > 40) return NNHIP_E_INVALID;
  for (int k = 0; k < count; ++k) {
    L.src[k] = src[k];
    L.dst[k] = dst[k];
  }
  transpose128_kernel<<<dim3(NF / 32, NF / 32, count), 256, 0, s>>>(L);
  LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Stand-alone building blocks of the C ABI (scatter-sum over the receiver, row gathers: newtonnet.py:210-226), each other's adjoints.
// They are linear maps and each other's adjoints, so autograd can differentiate through them twice
// (force-loss training: output.py:66-73 with create_graph=True, trainer.py:307-309).
//   segment_sum:  out[i][:] = sum_{e in [row_ptr[i], row_ptr[i+1])} x[e][:]      (deterministic scatter_sum)
//   gather_rows:  out[e][:] = x[idx[e]][:]
// One wave per output row; width (floats per row) must be a multiple of 2.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
segment_sum_kernel(const float* __restrict__ x, const int* __restrict__ row_ptr, int n_rows, int width,
                   float* __restrict__ out) {
  const int i = blockIdx.x * ROWS_PER_BLOCK + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (i >= n_rows) return;
  const int lane = threadIdx.x & 63;
  const int beg = row_ptr[i], end = row_ptr[i + 1];
  for (int c = 2 * lane; c < width; c += 128) {
    float2 acc = make_float2(0.f, 0.f);
    for (int e = beg; e < end; ++e) acc = acc + ld2(x + (size_t)e * width + c);
    st2(out + (size_t)i * width + c, acc);
  }
}

__global__ void __launch_bounds__(256)
gather_rows_kernel(const float* __restrict__ x, const int* __restrict__ idx, int n_out, int width,
                   float* __restrict__ out) {
  const int e = blockIdx.x * ROWS_PER_BLOCK + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (e >= n_out) return;
  const int lane = threadIdx.x & 63;
  const int j = idx[e];
  for (int c = 2 * lane; c < width; c += 128) st2(out + (size_t)e * width + c, ld2(x + (size_t)j * width + c));
}

extern "C" int nnhip_segment_sum(const float* x, const int32_t* row_ptr, int32_t n_rows, int32_t width, float* out,
                                 void* stream) {
  if (!row_ptr || !out || n_rows < 0 || width < 2 || (width & 1)) {
    nnhip_set_error("nnhip_segment_sum: bad arguments (width must be even)");
    return NNHIP_E_INVALID;
  }
  if (n_rows == 0) return NNHIP_OK;
  ScopedTimer t0(TC_EDGE, (hipStream_t)stream);
  segment_sum_kernel<<<cdiv(n_rows, ROWS_PER_BLOCK), 256, 0, (hipStream_t)stream>>>(x, row_ptr, n_rows, width, out);
  LAUNCH_CHECK();
  return NNHIP_OK;
}

extern "C" int nnhip_gather_rows(const float* x, const int32_t* idx, int32_t n_out, int32_t width, float* out,
                                 void* stream) {
  if (!idx || !out || n_out < 0 || width < 2 || (width & 1)) {
    nnhip_set_error("nnhip_gather_rows: bad arguments (width must be even)");
    return NNHIP_E_INVALID;
  }
  if (n_out == 0) return NNHIP_OK;
  ScopedTimer t0(TC_EDGE, (hipStream_t)stream);
  gather_rows_kernel<<<cdiv(n_out, ROWS_PER_BLOCK), 256, 0, (hipStream_t)stream>>>(x, idx, n_out, width, out);
  LAUNCH_CHECK();
  return NNHIP_OK;
}
