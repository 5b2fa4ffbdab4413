// Training kernels (gfx950, fp32): parameter gradients of any loss L(E, F), F = -dE/dpos, WITHOUT running autograd through the
// force graph.
//
// The reference trains by back-propagating through torch.autograd.grad(energy, pos, create_graph=True)
// (newtonnet/models/output.py:66-73; newtonnet/train/trainer.py:299-313; loss newtonnet/train/loss.py:48,72,96).  With
// c_b = dL/dE_b and d = dL/dF,
//     dL/dtheta = sum_b c_b dE_b/dtheta - D_d[grad_theta E_tot]
// (D_d: directional derivative along d in position space), i.e. the reverse sweep that yields grad_theta E is differentiated
// once more in FORWARD (tangent) mode along v = -d, with the reverse seed carried as the dual number 1 + eps c_b.  Four sweeps:
//   1 forward (values) and 2 reverse (values -> forces): the inference kernels (edge.hip / mlp128.hip / node128.hip), run stage
//     by stage with every intermediate kept;
//   3 tangent forward and 4 tangent reverse: the kernels in this file -- each is the derivative of one inference kernel along
//     the tangent direction: same mapping (one wave per receiver row, two edges per instruction, pair-space rows written by
//     the lower endpoint, no float atomics), reading values and tangents side by side;
//   then every weight gradient is a sum of two products  dW = A1^T B1 + A2^T B2  over the rows (pairs or atoms):
//     wgrad_kernel, a split-K fp32-MFMA reduction with deterministic slab + tree-free final sum.
// tests/tangent_ref.py writes the same four sweeps in fp64 torch; tests/test_hip_train.py checks every stage against it and
// the result against the oracle's autograd double backward.
#include <string.h>

#include "edge_common.h"

// ---------------------------------------------------------------------------------------------
// tangent of the edge geometry along v (per directed edge e = (i, j), d = pos_i - pos_j - shift):
//   dd = v_i - v_j;  dr = u . dd;  du = (dd - u dr) / r;  dx = dr / r_c          -> tgeo[e] = (du, dx)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
edge_tan_geom_kernel(const float* __restrict__ v, float sign, const int64_t* __restrict__ edge_index,
                     const float* __restrict__ geo, int n_edges, float inv_rc, float* __restrict__ tgeo) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_edges) return;
  const long i = edge_index[e], j = edge_index[(long)n_edges + e];
  const float4 g = reinterpret_cast<const float4*>(geo)[e];
  const float dx = sign * (v[3 * i] - v[3 * j]), dy = sign * (v[3 * i + 1] - v[3 * j + 1]), dz = sign * (v[3 * i + 2] - v[3 * j + 2]);
  const float dr = g.x * dx + g.y * dy + g.z * dz;
  const float ir = 1.0f / g.w;
  reinterpret_cast<float4*>(tgeo)[e] = make_float4((dx - g.x * dr) * ir, (dy - g.y * dr) * ir, (dz - g.z * dr) * ir, dr * inv_rc);
}

// ---------------------------------------------------------------------------------------------
// tangent of msg_fwd (edge.hip):   dmsg[p] = deps dx m_i m_j + eps (dm_i m_j + m_i dm_j);   da_mid[i] = da_in[i] + sum_e dmsg
// HAS_DM = false for the first layer (its m comes from the embedding: dm = 0, da_in = 0).
// ---------------------------------------------------------------------------------------------
template <bool HAS_DM>
__global__ void __launch_bounds__(64 * EDGE_ROWS)
msg_tan_fwd_kernel(const float* __restrict__ m, const float* __restrict__ dm, const int2* __restrict__ xg,
                   const float* __restrict__ tgeo, const float* __restrict__ table, const int* __restrict__ row_ptr,
                   const int* __restrict__ col, const int* __restrict__ pid, const float* __restrict__ da_in,
                   float* __restrict__ dmsg /*[P][F]*/, float* __restrict__ da_mid, int n_atoms) {
  const int i = wave_row(gridDim.x);
  if (i >= n_atoms) return;
  const int lane = threadIdx.x & 63;
  const int c4 = 4 * (lane & 31);
  const bool hi = lane >= 32;
  const float4 mi = ld4(m + (size_t)i * NF + c4);
  float4 dmi = make_float4(0.f, 0.f, 0.f, 0.f);
  if (HAS_DM) dmi = ld4(dm + (size_t)i * NF + c4);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const int beg = row_ptr[i], end = row_ptr[i + 1];
  for (int e = beg; e < end; e += 2) {
    const int e1 = min(e + 1, end - 1);
    const int j0 = col[e], j1 = col[e1];
    const int2 gx0 = xg[e], gx1 = xg[e1];
    const float dx0 = tgeo[4 * (size_t)e + 3], dx1 = tgeo[4 * (size_t)e1 + 3];
    const int jj = hi ? j1 : j0;
    const int2 gx = hi ? gx1 : gx0;
    const float dxe = hi ? dx1 : dx0;
    if (!hi || e + 1 < end) {
      const FilterW fw = filter_weights(__int_as_float(gx.y));
      float4 eps, deps;
      filter_value_deriv(table, gx.x, c4, fw, eps, deps);
      const float4 mj = ld4(m + (size_t)jj * NF + c4);
      float4 v = mul4(mul4(mul4(deps, dxe), mi), mj);
      if (HAS_DM) {
        const float4 dmj = ld4(dm + (size_t)jj * NF + c4);
        v = fma4(eps, fma4(dmi, mj, mul4(mi, dmj)), v);
      }
      if (jj > i) {
        const int p0 = pid[e], p1 = pid[e1];
        st4(dmsg + (size_t)(hi ? p1 : p0) * NF + c4, v);
      }
      acc = add4(acc, v);
    }
  }
  acc = add4(acc, upper_half(acc));
  if (!hi) {
    if (HAS_DM) acc = add4(acc, ld4(da_in + (size_t)i * NF + c4));
    st4(da_mid + (size_t)i * NF + c4, acc);
  }
}

// ---------------------------------------------------------------------------------------------
// tangent of force_fwd:
//   df_out[i][k] = df_in[i][k] + sum_e ( dphi1[p] u_e[k] + phi1[p] du_e[k] + dphi2[p] f_in[j][k] + phi2[p] df_in[j][k] )
// ---------------------------------------------------------------------------------------------
template <bool HAS_F>
__global__ void __launch_bounds__(64 * EDGE_ROWS)
force_tan_fwd_kernel(const float* __restrict__ phi1, const float* __restrict__ dphi1, const float* __restrict__ phi2,
                     const float* __restrict__ dphi2, const float* __restrict__ geo, const float* __restrict__ tgeo,
                     const int2* __restrict__ xg, const int* __restrict__ row_ptr, const int* __restrict__ col,
                     const int* __restrict__ pid, const float* __restrict__ f_in, const float* __restrict__ df_in,
                     float* __restrict__ df_out, int n_atoms) {
  const int i = wave_row(gridDim.x);
  if (i >= n_atoms) return;
  const int lane = threadIdx.x & 63;
  const int c4 = 4 * (lane & 31);
  const bool hi = lane >= 32;
  float4 acc[3];
#pragma unroll
  for (int k = 0; k < 3; ++k)
    acc[k] = (HAS_F && !hi) ? ld4(df_in + ((size_t)i * 3 + k) * NF + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
  const int beg = row_ptr[i], end = row_ptr[i + 1];
  for (int e = beg; e < end; e += 2) {
    const int e1 = min(e + 1, end - 1);
    const float4 g0 = reinterpret_cast<const float4*>(geo)[e], g1 = reinterpret_cast<const float4*>(geo)[e1];
    const float4 t0 = reinterpret_cast<const float4*>(tgeo)[e], t1 = reinterpret_cast<const float4*>(tgeo)[e1];
    const int p0 = pid[e], p1 = pid[e1];
    const int gz0 = xg[e].x, gz1 = xg[e1].x;
    const float4 g = hi ? g1 : g0, tg = hi ? t1 : t0;
    const size_t p = (size_t)(hi ? p1 : p0);
    if ((!hi || e + 1 < end) && (hi ? gz1 : gz0) != FT_ZERO_ROW) {
      const float4 v1 = ld4(phi1 + p * NF + c4), dv1 = ld4(dphi1 + p * NF + c4);
      acc[0] = fma4(dv1, g.x, fma4(v1, tg.x, acc[0]));
      acc[1] = fma4(dv1, g.y, fma4(v1, tg.y, acc[1]));
      acc[2] = fma4(dv1, g.z, fma4(v1, tg.z, acc[2]));
      if (HAS_F) {
        const int j0 = col[e], j1 = col[e1];
        const int j = hi ? j1 : j0;
        const float4 v2 = ld4(phi2 + p * NF + c4), dv2 = ld4(dphi2 + p * NF + c4);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          acc[k] = fma4(dv2, ld4(f_in + ((size_t)j * 3 + k) * NF + c4), acc[k]);
          acc[k] = fma4(v2, ld4(df_in + ((size_t)j * 3 + k) * NF + c4), acc[k]);
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float4 o = add4(acc[k], upper_half(acc[k]));
    if (!hi) st4(df_out + ((size_t)i * 3 + k) * NF + c4, o);
  }
}

// ---------------------------------------------------------------------------------------------
// tangent of force_bwd (no g_u: the tangent of the force itself is never needed, only what feeds weight gradients):
//   dg_phi1[p] = sum_k (dgf_i - dgf_j)[k] u_e[k] + (gf_i - gf_j)[k] du_e[k]                        -> dg_h12[p][0:F]
//   dg_phi2[p] = sum_k dgf_i[k] f_j[k] + gf_i[k] df_j[k] + dgf_j[k] f_i[k] + gf_j[k] df_i[k]       -> dg_h12[p][F:2F]
//   dG_fin[i][k] = dgf[i][k] + sum_{e in row i} dphi2[p] gf[j][k] + phi2[p] dgf[j][k]
// ---------------------------------------------------------------------------------------------
template <bool HAS_F>
__global__ void __launch_bounds__(64 * EDGE_ROWS)
force_tan_bwd_kernel(const float* __restrict__ gf, const float* __restrict__ dgf, const float* __restrict__ phi2,
                     const float* __restrict__ dphi2, const float* __restrict__ geo, const float* __restrict__ tgeo,
                     const int2* __restrict__ xg, const int* __restrict__ row_ptr, const int* __restrict__ col,
                     const int* __restrict__ pid, const float* __restrict__ f_in, const float* __restrict__ df_in,
                     float* __restrict__ dg_h12 /*[P][2F]*/, float* __restrict__ dg_fin, int n_atoms) {
  const int i = wave_row(gridDim.x);
  if (i >= n_atoms) return;
  const int lane = threadIdx.x & 63;
  const int c4 = 4 * (lane & 31);
  const bool hi = lane >= 32;
  float4 gfi[3], dgfi[3], fi[3], dfi[3], acc[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    gfi[k] = ld4(gf + ((size_t)i * 3 + k) * NF + c4);
    dgfi[k] = ld4(dgf + ((size_t)i * 3 + k) * NF + c4);
    acc[k] = hi ? make_float4(0.f, 0.f, 0.f, 0.f) : dgfi[k];
    if (HAS_F) {
      fi[k] = ld4(f_in + ((size_t)i * 3 + k) * NF + c4);
      dfi[k] = ld4(df_in + ((size_t)i * 3 + k) * NF + c4);
    }
  }
  const int beg = row_ptr[i], end = row_ptr[i + 1];
  const int mid = row_mid(col, beg, end, i, lane);
  if (HAS_F) {   // [beg, mid): pairs owned by the other endpoint -- the phi2 gather only
    for (int e = beg; e < mid; e += 2) {
      const int e1 = min(e + 1, mid - 1);
      const int p0 = pid[e], p1 = pid[e1];
      const int j0 = col[e], j1 = col[e1];
      const int gz0 = xg[e].x, gz1 = xg[e1].x;
      const size_t p = (size_t)(hi ? p1 : p0);
      const int j = hi ? j1 : j0;
      if ((!hi || e + 1 < mid) && (hi ? gz1 : gz0) != FT_ZERO_ROW) {
        const float4 v2 = ld4(phi2 + p * NF + c4), dv2 = ld4(dphi2 + p * NF + c4);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          acc[k] = fma4(dv2, ld4(gf + ((size_t)j * 3 + k) * NF + c4), acc[k]);
          acc[k] = fma4(v2, ld4(dgf + ((size_t)j * 3 + k) * NF + c4), acc[k]);
        }
      }
    }
  }
  for (int e = mid; e < end; e += 2) {
    const int e1 = min(e + 1, end - 1);
    const int p0 = pid[e], p1 = pid[e1];
    const int j0 = col[e], j1 = col[e1];
    const float4 g0 = reinterpret_cast<const float4*>(geo)[e], g1 = reinterpret_cast<const float4*>(geo)[e1];
    const float4 t0 = reinterpret_cast<const float4*>(tgeo)[e], t1 = reinterpret_cast<const float4*>(tgeo)[e1];
    const int gz0 = xg[e].x, gz1 = xg[e1].x;
    const size_t p = (size_t)(hi ? p1 : p0);
    const int j = hi ? j1 : j0;
    const float4 g = hi ? g1 : g0, tg = hi ? t1 : t0;
    const bool inside = (hi ? gz1 : gz0) != FT_ZERO_ROW;
    if ((!hi || e + 1 < end) && !inside) {
      const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
      st4(dg_h12 + p * 2 * NF + c4, zero);
      if (HAS_F) st4(dg_h12 + p * 2 * NF + NF + c4, zero);
    }
    if ((!hi || e + 1 < end) && inside) {
      float4 gfj[3], dgfj[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        gfj[k] = ld4(gf + ((size_t)j * 3 + k) * NF + c4);
        dgfj[k] = ld4(dgf + ((size_t)j * 3 + k) * NF + c4);
      }
      float4 gp1 = mul4(sub4(dgfi[0], dgfj[0]), g.x);
      gp1 = fma4(sub4(dgfi[1], dgfj[1]), g.y, gp1);
      gp1 = fma4(sub4(dgfi[2], dgfj[2]), g.z, gp1);
      gp1 = fma4(sub4(gfi[0], gfj[0]), tg.x, gp1);
      gp1 = fma4(sub4(gfi[1], gfj[1]), tg.y, gp1);
      gp1 = fma4(sub4(gfi[2], gfj[2]), tg.z, gp1);
      st4(dg_h12 + p * 2 * NF + c4, gp1);
      if (HAS_F) {
        const float4 v2 = ld4(phi2 + p * NF + c4), dv2 = ld4(dphi2 + p * NF + c4);
        float4 gp2 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          acc[k] = fma4(dv2, gfj[k], acc[k]);
          acc[k] = fma4(v2, dgfj[k], acc[k]);
          gp2 = fma4(dgfi[k], ld4(f_in + ((size_t)j * 3 + k) * NF + c4), gp2);
          gp2 = fma4(gfi[k], ld4(df_in + ((size_t)j * 3 + k) * NF + c4), gp2);
          gp2 = fma4(dgfj[k], fi[k], gp2);
          gp2 = fma4(gfj[k], dfi[k], gp2);
        }
        st4(dg_h12 + p * 2 * NF + NF + c4, gp2);
      }
    }
  }
  if (HAS_F) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float4 o = add4(acc[k], upper_half(acc[k]));
      if (!hi) st4(dg_fin + ((size_t)i * 3 + k) * NF + c4, o);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// tangent of msg_bwd, plus the radial-filter adjoint rows that feed the message_edgepart weight gradient:
//   G = g_msg[p] + GA_i + GA_j;   dG = dg_msg[p] + dGA_i + dGA_j
//   dg_m[i]   = sum_{e in row i} dG eps m_j + G (deps dx) m_j + G eps dm_j
//   g_eps[p]  = G m_i m_j;   dg_eps[p] = dG m_i m_j + G (dm_i m_j + m_i dm_j)          (written by the pair's owner)
// HAS_DM = false for the first layer (dm = 0).
// ---------------------------------------------------------------------------------------------
template <bool HAS_DM>
__global__ void __launch_bounds__(64 * EDGE_ROWS)
msg_tan_bwd_kernel(const float* __restrict__ g_msg, const float* __restrict__ dg_msg, const float* __restrict__ ga,
                   const float* __restrict__ dga, const float* __restrict__ m, const float* __restrict__ dm,
                   const int2* __restrict__ xg, const float* __restrict__ tgeo, const float* __restrict__ table,
                   const int* __restrict__ row_ptr, const int* __restrict__ col, const int* __restrict__ pid,
                   float* __restrict__ dg_m, float* __restrict__ g_eps /*[P][F]*/, float* __restrict__ dg_eps, int n_atoms) {
  const int i = wave_row(gridDim.x);
  if (i >= n_atoms) return;
  const int lane = threadIdx.x & 63;
  const int c4 = 4 * (lane & 31);
  const bool hi = lane >= 32;
  const float4 mi = ld4(m + (size_t)i * NF + c4);
  const float4 gai = ld4(ga + (size_t)i * NF + c4), dgai = ld4(dga + (size_t)i * NF + c4);
  float4 dmi = make_float4(0.f, 0.f, 0.f, 0.f);
  if (HAS_DM) dmi = ld4(dm + (size_t)i * NF + c4);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const int beg = row_ptr[i], end = row_ptr[i + 1];
  for (int e = beg; e < end; e += 2) {
    const int e1 = min(e + 1, end - 1);
    const int j0 = col[e], j1 = col[e1];
    const int p0 = pid[e], p1 = pid[e1];
    const int2 gx0 = xg[e], gx1 = xg[e1];
    const float dx0 = tgeo[4 * (size_t)e + 3], dx1 = tgeo[4 * (size_t)e1 + 3];
    const int j = hi ? j1 : j0;
    const size_t p = (size_t)(hi ? p1 : p0);
    const int2 gx = hi ? gx1 : gx0;
    const float dxe = hi ? dx1 : dx0;
    if (!hi || e + 1 < end) {
      const float4 mj = ld4(m + (size_t)j * NF + c4);
      const float4 G = add4(add4(ld4(g_msg + p * NF + c4), gai), ld4(ga + (size_t)j * NF + c4));
      const float4 dG = add4(add4(ld4(dg_msg + p * NF + c4), dgai), ld4(dga + (size_t)j * NF + c4));
      const FilterW fw = filter_weights(__int_as_float(gx.y));
      float4 eps, deps;
      filter_value_deriv(table, gx.x, c4, fw, eps, deps);
      float4 dmj = make_float4(0.f, 0.f, 0.f, 0.f);
      if (HAS_DM) dmj = ld4(dm + (size_t)j * NF + c4);
      // (dG eps + G deps dx) m_j + G eps dm_j
      float4 t = fma4(dG, eps, mul4(mul4(G, deps), dxe));
      acc = fma4(t, mj, acc);
      if (HAS_DM) acc = fma4(mul4(G, eps), dmj, acc);
      if (j > i) {
        const float4 mm = mul4(mi, mj);
        st4(g_eps + p * NF + c4, mul4(G, mm));
        float4 de = mul4(dG, mm);
        if (HAS_DM) de = fma4(G, fma4(dmi, mj, mul4(mi, dmj)), de);
        st4(dg_eps + p * NF + c4, de);
      }
    }
  }
  acc = add4(acc, upper_half(acc));
  if (!hi) st4(dg_m + (size_t)i * NF + c4, acc);
}

// ---------------------------------------------------------------------------------------------
// node-level elementwise pieces of the tangent sweeps ([N][F] / [N][3][F] rows, float4 per thread)
// ---------------------------------------------------------------------------------------------
// tangent of the energy update (newtonnet.py:229-231):  da_out = da_mid + sum_k (df_k q_k + f_k dq_k)
__global__ void __launch_bounds__(256)
update_tan_fwd_kernel(const float* __restrict__ da_mid, const float* __restrict__ f, const float* __restrict__ df,
                      const float* __restrict__ q, const float* __restrict__ dq, int n_atoms, float* __restrict__ da_out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)n_atoms * (NF / 4)) return;
  const size_t i = t / (NF / 4), c = t % (NF / 4);
  float4 acc = reinterpret_cast<const float4*>(da_mid)[t];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const size_t o = (i * 3 + k) * (NF / 4) + c;
    acc = fma4(reinterpret_cast<const float4*>(df)[o], reinterpret_cast<const float4*>(q)[o], acc);
    acc = fma4(reinterpret_cast<const float4*>(f)[o], reinterpret_cast<const float4*>(dq)[o], acc);
  }
  reinterpret_cast<float4*>(da_out)[t] = acc;
}

// tangent of the update adjoint, first half (the product with W_u follows as an accumulating linear):
//   gq_k = GA f_k;   dgq_k = dGA f_k + GA df_k;   dgf_k = dG_f,k + dGA q_k + GA dq_k          (dG_f may be NULL = 0)
__global__ void __launch_bounds__(256)
update_tan_bwd_kernel(const float* __restrict__ ga, const float* __restrict__ dga, const float* __restrict__ f,
                      const float* __restrict__ df, const float* __restrict__ q, const float* __restrict__ dq,
                      const float* __restrict__ dgf_in, int n_atoms, float* __restrict__ gq, float* __restrict__ dgq,
                      float* __restrict__ dgf) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)n_atoms * (NF / 4)) return;
  const size_t i = t / (NF / 4), c = t % (NF / 4);
  const float4 a = reinterpret_cast<const float4*>(ga)[t], da = reinterpret_cast<const float4*>(dga)[t];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const size_t o = (i * 3 + k) * (NF / 4) + c;
    const float4 fk = reinterpret_cast<const float4*>(f)[o], dfk = reinterpret_cast<const float4*>(df)[o];
    reinterpret_cast<float4*>(gq)[o] = mul4(a, fk);
    reinterpret_cast<float4*>(dgq)[o] = fma4(da, fk, mul4(a, dfk));
    float4 o4 = fma4(da, reinterpret_cast<const float4*>(q)[o], mul4(a, reinterpret_cast<const float4*>(dq)[o]));
    if (dgf_in) o4 = add4(o4, reinterpret_cast<const float4*>(dgf_in)[o]);
    reinterpret_cast<float4*>(dgf)[o] = o4;
  }
}

// Seed of the tangent reverse sweep at the energy head (output.py:98-100, scalers.py:55-58), one wave per atom.
// With the reverse seed 1 + eps c_b (c = dL/dE of the atom's molecule):
//   dg_e2[i]   = sc w4 (c act'(e2) + act''(e2) de2)
//   w4row[i]   = sc (c act(e2) + act'(e2) de2)                          column sums -> dL/d w4
//   scal[i]    = (c eps_i + deps_i,  c,  sc c, 0)                       species sums -> dL/d scale[z], dL/d shift[z]; sum -> b4
// where eps_i = <act(e2), w4> + b4 and deps_i = <act'(e2) de2, w4>.
__global__ void __launch_bounds__(256)
head_seed_tan_kernel(const float* __restrict__ e2, const float* __restrict__ de2, const float* __restrict__ w4,
                     const float* __restrict__ b4, const float* __restrict__ scale, const int64_t* __restrict__ z,
                     const int64_t* __restrict__ batch, const float* __restrict__ g_energy, int n_atoms, int act,
                     float* __restrict__ dg_e2, float* __restrict__ w4row, float* __restrict__ scal /*[N][4]*/) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n_atoms) return;
  const int lane = threadIdx.x & 63;
  const float2 h = ld2(e2 + (size_t)i * NF + 2 * lane), dh = ld2(de2 + (size_t)i * NF + 2 * lane);
  const float2 w = ld2(w4 + 2 * lane);
  const float c = g_energy[batch[i]];
  const float sc = scale ? scale[z[i]] : 1.0f;
  const float ax = act_any(h.x, act), ay = act_any(h.y, act);
  const float bx = dact_any(h.x, act), by = dact_any(h.y, act);
  const float cx = d2act_any(h.x, act), cy = d2act_any(h.y, act);
  st2(dg_e2 + (size_t)i * NF + 2 * lane,
      make_float2(sc * w.x * fmaf(c, bx, cx * dh.x), sc * w.y * fmaf(c, by, cy * dh.y)));
  st2(w4row + (size_t)i * NF + 2 * lane, make_float2(sc * fmaf(c, ax, bx * dh.x), sc * fmaf(c, ay, by * dh.y)));
  const float eps = wave_sum(fmaf(ax, w.x, ay * w.y)) + b4[0];
  const float deps = wave_sum(fmaf(bx * dh.x, w.x, by * dh.y * w.y));
  if (lane == 0) reinterpret_cast<float4*>(scal)[i] = make_float4(fmaf(c, eps, deps), c, sc * c, 0.f);
}

// Per-pair radial-basis rows for the message_edgepart weight gradient, written by the pair's owner edge (i < j):
//   rb[p][0:nb] = rbf_e,   rb[p][32 + 0:nb] = drbf_e dx_e          ([P][64] rows, zero padded)
__global__ void __launch_bounds__(256)
pair_rbf_kernel(const float* __restrict__ rbf, const float* __restrict__ drbf, const float* __restrict__ tgeo,
                const int64_t* __restrict__ edge_index, const int* __restrict__ pid, int n_edges, int nb,
                float* __restrict__ rb /*[P][64]*/) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t e = t >> 5;
  const int n = (int)(t & 31);
  if (e >= (size_t)n_edges) return;
  if (edge_index[e] > edge_index[(size_t)n_edges + e]) return;   // the lower endpoint's edge owns the pair
  const size_t p = (size_t)pid[e];
  const float dx = tgeo[4 * e + 3];
  rb[p * 64 + n] = n < nb ? rbf[e * nb + n] : 0.f;
  rb[p * 64 + 32 + n] = n < nb ? drbf[e * nb + n] * dx : 0.f;
}

// ---------------------------------------------------------------------------------------------
// Weight gradients:  dW[o][i] = sum_rows A1[r][o] B1[r][i] + A2[r][o] B2[r][i]      (M rows: pairs, atoms or 3 x atoms)
// Split-K over the rows on the fp32 matrix cores.  v_mfma_f32_32x32x2_f32 contracts over k = 2 rows per instruction: lane
// (c, h) feeds row r0 + h.  Its 32 "A rows" / "B columns" are mapped to features 4c + w: a lane then loads 16 contiguous
// bytes of a row (features 4c .. 4c+3) and the four components are the four 32-feature classes -- wave w of the workgroup
// owns output-feature class w (4 accumulators, one per input-feature class).  Every workgroup reduces its own slice of rows
// into a private [128][128] slab; wgrad_reduce_kernel sums the slabs in a fixed order (deterministic, no atomics).
// Operand forms (prologue of the loads), by problem type:
//   WG_PLAIN   A1, B1, A2, B2 as stored
//   WG_ACT     B1 = act(hB),  B2 = act'(hB) dhB            (A1 = dg_out, A2 = g_out: second-linear weights)
//   WG_TDACT   A2 = tA act'(hA)                            (A1 = dg_hidden, B1 = x, B2 = dx: first-linear weights)
// A2 / B2 may be NULL (single product).  NB = 32: B rows are 32 wide (the radial basis), one accumulator per wave.
// ---------------------------------------------------------------------------------------------
enum { WG_PLAIN = 0, WG_ACT = 1, WG_TDACT = 2 };
typedef nnhip_wgrad_problem WgProb;   // the problem table lives in device memory (uploaded once per workspace by the caller)

__device__ __forceinline__ float comp(const float4& v, int w) { return w == 0 ? v.x : (w == 1 ? v.y : (w == 2 ? v.z : v.w)); }

// Workgroup = 4 waves; a trip handles WG_R rows.  All 256 threads fetch the trip's operand rows once (float4 per lane),
// apply the prologue (activation factors are evaluated once per element, not once per wave) and stage them in LDS:
// B rows as stored ([row][128]), A rows split by output-feature class ([class][row][32]) so that wave w reads its class
// with unit stride.  Two LDS buffers: the next trip's rows are requested before this trip's MFMAs and written behind them.
#define WG_R 16
#ifndef WG_WAVES
#define WG_WAVES 4   // 4: wave = output class, 4 column blocks each; 8 (tooling): two waves per class, 2 column blocks each
#endif
#define WG_THREADS (64 * WG_WAVES)
#define WG_Q (WG_R * 32 / WG_THREADS)   // rows a thread fetches per trip
#define WG_NACC (16 / WG_WAVES)         // accumulator blocks per wave
struct WgStage {   // what one thread fetched for the next trip: rows (t >> 5) + (WG_THREADS / 32) q, lane column c = t & 31
  float4 a1[WG_Q], a2[WG_Q], b1[WG_Q], b2[WG_Q], ha[WG_Q];   // raw as loaded (b1 / b2 hold hB / dhB for WG_ACT); wg_finish applies the prologue
  bool live[WG_Q];
};
struct WgLds {
  float a1[4][WG_R][32], a2[4][WG_R][32];
  float b1[WG_R][NF], b2[WG_R][NF];
};

// Issue the loads only: nothing here consumes a loaded value, so the requests stay in flight under the MFMAs that follow.
// (TYPE / TWO / NB32 are compile-time: the trip loop of each operand form is straight-line code.)
template <int TYPE, bool TWO, bool NB32>
__device__ __forceinline__ void wg_fetch(const WgProb& P, int r0, int r_end, int r_safe, WgStage& g) {
  const int c = threadIdx.x & 31;
#pragma unroll
  for (int q = 0; q < WG_Q; ++q) {
    const int r = r0 + (threadIdx.x >> 5) + (WG_THREADS / 32) * q;
    g.live[q] = r < r_end;
    const int rr = g.live[q] ? r : r_safe;
    g.a1[q] = ld4(P.A1 + (size_t)rr * P.lda1 + 4 * c);
    if (NB32) {   // B rows are 32 floats wide: lane column c carries one of them (in .x)
      g.b1[q].x = P.B1[(size_t)rr * P.ldb1 + c];
      if (TWO) g.b2[q].x = P.B2[(size_t)rr * P.ldb2 + c];
    } else if (TYPE == WG_ACT) {
      g.b1[q] = ld4(P.hB + (size_t)rr * P.ldh + 4 * c);
      if (TWO) g.b2[q] = ld4(P.dhB + (size_t)rr * P.ldh + 4 * c);
    } else {
      g.b1[q] = ld4(P.B1 + (size_t)rr * P.ldb1 + 4 * c);
      if (TWO) g.b2[q] = ld4(P.B2 + (size_t)rr * P.ldb2 + 4 * c);
    }
    if (TWO) {
      g.a2[q] = ld4(P.A2 + (size_t)rr * P.lda2 + 4 * c);
      if (TYPE == WG_TDACT) g.ha[q] = ld4(P.hA + (size_t)rr * P.ldh + 4 * c);
    }
  }
}
// the prologue of the operand forms, applied once per element (after the MFMAs of the current trip)
template <int TYPE, bool TWO, bool NB32>
__device__ __forceinline__ void wg_finish(const WgProb& P, WgStage& g) {
  const int act = P.activation;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int q = 0; q < WG_Q; ++q) {
    if (!NB32 && TYPE == WG_ACT) {
      const float4 hv = g.b1[q], dh = g.b2[q];
      g.b1[q] = make_float4(act_any(hv.x, act), act_any(hv.y, act), act_any(hv.z, act), act_any(hv.w, act));
      if (TWO)
        g.b2[q] = make_float4(dact_any(hv.x, act) * dh.x, dact_any(hv.y, act) * dh.y, dact_any(hv.z, act) * dh.z,
                              dact_any(hv.w, act) * dh.w);
    }
    if (TWO && TYPE == WG_TDACT) {
      const float4 hv = g.ha[q];
      g.a2[q] = make_float4(g.a2[q].x * dact_any(hv.x, act), g.a2[q].y * dact_any(hv.y, act), g.a2[q].z * dact_any(hv.z, act),
                            g.a2[q].w * dact_any(hv.w, act));
    }
    if (!g.live[q]) {
      g.a1[q] = zero;
      if (TWO) g.a2[q] = zero;
    }
  }
}
template <bool TWO, bool NB32>
__device__ __forceinline__ void wg_commit(WgLds& L, const WgStage& g) {
  const int c = threadIdx.x & 31;
#pragma unroll
  for (int q = 0; q < WG_Q; ++q) {
    const int row = (threadIdx.x >> 5) + (WG_THREADS / 32) * q;
    L.a1[0][row][c] = g.a1[q].x;
    L.a1[1][row][c] = g.a1[q].y;
    L.a1[2][row][c] = g.a1[q].z;
    L.a1[3][row][c] = g.a1[q].w;
    if (NB32)
      L.b1[row][c] = g.b1[q].x;
    else
      *reinterpret_cast<float4*>(&L.b1[row][4 * c]) = g.b1[q];
    if (TWO) {
      L.a2[0][row][c] = g.a2[q].x;
      L.a2[1][row][c] = g.a2[q].y;
      L.a2[2][row][c] = g.a2[q].z;
      L.a2[3][row][c] = g.a2[q].w;
      if (NB32)
        L.b2[row][c] = g.b2[q].x;
      else
        *reinterpret_cast<float4*>(&L.b2[row][4 * c]) = g.b2[q];
    }
  }
}

template <int TYPE, bool TWO, bool NB32>
__device__ __forceinline__ void wg_run(const WgProb& P, WgLds* lds, int r_beg, int r_end, int w, int c, int h, f32x16 (&acc)[WG_NACC]) {
  WgStage g;
  wg_fetch<TYPE, TWO, NB32>(P, r_beg, r_end, r_beg, g);
  wg_finish<TYPE, TWO, NB32>(P, g);
  wg_commit<TWO, NB32>(lds[0], g);
  __syncthreads();
  int buf = 0;
  for (int r0 = r_beg; r0 < r_end; r0 += WG_R) {
    const bool more = r0 + WG_R < r_end;
    if (more) wg_fetch<TYPE, TWO, NB32>(P, r0 + WG_R, r_end, r_beg, g);   // requests in flight under the MFMAs below
    __builtin_amdgcn_sched_barrier(0);          // (pin them here: the scheduler would sink the loads to their uses)
    const WgLds& L = lds[buf];
#pragma unroll
    for (int kk = 0; kk < WG_R / 2; ++kk) {
      const int row = 2 * kk + h;
      const float a1 = L.a1[w & 3][row][c];
      if (NB32) {
        if (WG_WAVES == 4 || w < 4) {   // (8-wave form: the second wave of a class idles on 32-column problems)
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, L.b1[row][c], acc[0], 0, 0, 0);
          if (TWO) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(L.a2[w & 3][row][c], L.b2[row][c], acc[0], 0, 0, 0);
        }
      } else {
#if WG_WAVES == 4
        const float4 b1 = *reinterpret_cast<const float4*>(&L.b1[row][4 * c]);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1.x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1.y, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1.z, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1.w, acc[3], 0, 0, 0);
        if (TWO) {
          const float a2 = L.a2[w][row][c];
          const float4 b2 = *reinterpret_cast<const float4*>(&L.b2[row][4 * c]);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b2.x, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b2.y, acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b2.z, acc[2], 0, 0, 0);
          acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b2.w, acc[3], 0, 0, 0);
        }
#else
        const int hf = 2 * (w >> 2);
        const float2 b1 = *reinterpret_cast<const float2*>(&L.b1[row][4 * c + hf]);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1.x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1.y, acc[1], 0, 0, 0);
        if (TWO) {
          const float a2 = L.a2[w & 3][row][c];
          const float2 b2 = *reinterpret_cast<const float2*>(&L.b2[row][4 * c + hf]);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b2.x, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b2.y, acc[1], 0, 0, 0);
        }
#endif
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (more) {
      wg_finish<TYPE, TWO, NB32>(P, g);
      wg_commit<TWO, NB32>(lds[buf ^ 1], g);
    }
    __syncthreads();
    buf ^= 1;
  }
}

// Rows of a problem per split-K workgroup, and how many workgroups of the (chunks x problems) grid the problem uses: `chunks` is
// sized for the LARGEST problem of the table; P.pad_ (set by the caller, 0 = off) is that problem's rows per chunk, a floor for
// every other one -- a node-level problem of N rows next to pair-level ones of 7 N rows then writes (and wgrad_reduce_kernel
// reads) a seventh of the slabs instead of all of them.
__device__ __forceinline__ void wg_partition(const WgProb& P, int M, int chunks, int trip, int& per, int& nch) {
  per = ((M + trip * chunks - 1) / (trip * chunks)) * trip;
  if (P.pad_ > 0) per = max(per, ((P.pad_ + trip - 1) / trip) * trip);
  nch = per > 0 ? (M + per - 1) / per : 0;
}

__global__ void __launch_bounds__(WG_THREADS)
wgrad_kernel(const WgProb* __restrict__ probs, int chunks, float* __restrict__ slabs, int pair_rows) {
  extern __shared__ __attribute__((aligned(16))) char wg_lds_raw[];
  WgLds* lds = reinterpret_cast<WgLds*>(wg_lds_raw);   // [2]
  WgProb P = probs[blockIdx.y];
  if (P.M < 0) P.M = pair_rows;                         // pair-level problem: the row count of THIS step (capacity-sized tables)
  if (!P.lda1) P.lda1 = NF;
  if (!P.lda2) P.lda2 = NF;
  if (!P.ldb1) P.ldb1 = NF;
  if (!P.ldb2) P.ldb2 = NF;
  if (!P.ldh) P.ldh = NF;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // output-feature class of this wave
  const int c = lane & 31, h = lane >> 5;
  const int M = P.M;
  // rows of this workgroup: contiguous, a multiple of the trip size
  int per, nch;
  wg_partition(P, M, chunks, WG_R, per, nch);
  if ((int)blockIdx.x >= nch) return;            // (uniform: this problem uses fewer workgroups than the grid has)
  const int r_beg = blockIdx.x * per, r_end = min(M, r_beg + per);
  f32x16 acc[WG_NACC];
#pragma unroll
  for (int q = 0; q < WG_NACC; ++q)
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[q][k] = 0.f;
  const bool two = P.A2 != nullptr;
  const bool nb32 = P.b_cols32 != 0;
  if (r_beg < r_end) {   // (uniform per workgroup: one straight-line trip loop per operand form)
    if (nb32) {
      if (two) wg_run<WG_PLAIN, true, true>(P, lds, r_beg, r_end, w, c, h, acc);
      else wg_run<WG_PLAIN, false, true>(P, lds, r_beg, r_end, w, c, h, acc);
    } else if (P.type == WG_ACT) {
      if (two) wg_run<WG_ACT, true, false>(P, lds, r_beg, r_end, w, c, h, acc);
      else wg_run<WG_ACT, false, false>(P, lds, r_beg, r_end, w, c, h, acc);
    } else if (P.type == WG_TDACT) {
      if (two) wg_run<WG_TDACT, true, false>(P, lds, r_beg, r_end, w, c, h, acc);
      else wg_run<WG_PLAIN, false, false>(P, lds, r_beg, r_end, w, c, h, acc);
    } else {
      if (two) wg_run<WG_PLAIN, true, false>(P, lds, r_beg, r_end, w, c, h, acc);
      else wg_run<WG_PLAIN, false, false>(P, lds, r_beg, r_end, w, c, h, acc);
    }
  }
  // D[row][col]: row = (k & 3) + 8 (k >> 2) + 4 h -> output feature 4 row + w; col = c -> input feature 4 c + q
  float* slab = slabs + ((size_t)blockIdx.y * chunks + blockIdx.x) * NF * NF;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int row = (k & 3) + 8 * (k >> 2) + 4 * h;
    const int o = 4 * row + (w & 3);
#if WG_WAVES == 4
    if (nb32)
      slab[(size_t)o * NF + c] = acc[0][k];     // input "feature" c = basis index
    else
      st4(slab + (size_t)o * NF + 4 * c, make_float4(acc[0][k], acc[1][k], acc[2][k], acc[3][k]));
#else
    if (nb32) {
      if (w < 4) slab[(size_t)o * NF + c] = acc[0][k];
    } else {
      *reinterpret_cast<float2*>(slab + (size_t)o * NF + 4 * c + 2 * (w >> 2)) = make_float2(acc[0][k], acc[1][k]);
    }
#endif
  }
}

// ---------------------------------------------------------------------------------------------
// bf16-operand form of the same products (BASELINE configs[2] names bf16): operands are rounded to bf16 AFTER their fp32
// prologue, the products accumulate in fp32 on v_mfma_f32_32x32x16_bf16 (16 rows per instruction, 16x the fp32 rate), so the
// kernel is bound by the operand traffic alone.  The MFMA wants 8 consecutive ROWS per lane for a fixed feature: the LDS
// image is transposed, [class][lane column][row] bf16 with a 40-element pitch (conflict-free ds_read_b128), two rows packed
// per 32-bit LDS write.  Trip = 32 rows; same slabs, same deterministic reduction.
// ---------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#define WB_R 32
#define WB_P 40                                  // row pitch (bf16 elements) of one (class, column) line
struct WbLds {
  unsigned short a1[4][32][WB_P], a2[4][32][WB_P], b1[4][32][WB_P], b2[4][32][WB_P];
};
struct WbStage {   // rows 2g, 2g+1, 2g+16, 2g+17 of the trip (g = t >> 5), lane column c = t & 31
  float4 a1[4], a2[4], b1[4], b2[4], ha[4];
  bool live[4];
};
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
  bf16x2 v;
  v[0] = (__bf16)lo;
  v[1] = (__bf16)hi;
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ int wb_row(int q) { return 2 * (threadIdx.x >> 5) + (q & 1) + 16 * (q >> 1); }

template <int TYPE, bool TWO, bool NB32>
__device__ __forceinline__ void wb_fetch(const WgProb& P, int r0, int r_end, int r_safe, WbStage& g) {
  const int c = threadIdx.x & 31;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = r0 + wb_row(q);
    g.live[q] = r < r_end;
    const int rr = g.live[q] ? r : r_safe;
    g.a1[q] = ld4(P.A1 + (size_t)rr * P.lda1 + 4 * c);
    if (NB32) {
      g.b1[q].x = P.B1[(size_t)rr * P.ldb1 + c];
      if (TWO) g.b2[q].x = P.B2[(size_t)rr * P.ldb2 + c];
    } else if (TYPE == WG_ACT) {
      g.b1[q] = ld4(P.hB + (size_t)rr * P.ldh + 4 * c);
      if (TWO) g.b2[q] = ld4(P.dhB + (size_t)rr * P.ldh + 4 * c);
    } else {
      g.b1[q] = ld4(P.B1 + (size_t)rr * P.ldb1 + 4 * c);
      if (TWO) g.b2[q] = ld4(P.B2 + (size_t)rr * P.ldb2 + 4 * c);
    }
    if (TWO) {
      g.a2[q] = ld4(P.A2 + (size_t)rr * P.lda2 + 4 * c);
      if (TYPE == WG_TDACT) g.ha[q] = ld4(P.hA + (size_t)rr * P.ldh + 4 * c);
    }
  }
}
template <int TYPE, bool TWO, bool NB32>
__device__ __forceinline__ void wb_finish_commit(const WgProb& P, WbLds& L, WbStage& g) {
  const int act = P.activation;
  const int c = threadIdx.x & 31;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (!NB32 && TYPE == WG_ACT) {
      const float4 hv = g.b1[q], dh = g.b2[q];
      g.b1[q] = make_float4(act_any(hv.x, act), act_any(hv.y, act), act_any(hv.z, act), act_any(hv.w, act));
      if (TWO)
        g.b2[q] = make_float4(dact_any(hv.x, act) * dh.x, dact_any(hv.y, act) * dh.y, dact_any(hv.z, act) * dh.z,
                              dact_any(hv.w, act) * dh.w);
    }
    if (TWO && TYPE == WG_TDACT) {
      const float4 hv = g.ha[q];
      g.a2[q] = make_float4(g.a2[q].x * dact_any(hv.x, act), g.a2[q].y * dact_any(hv.y, act), g.a2[q].z * dact_any(hv.z, act),
                            g.a2[q].w * dact_any(hv.w, act));
    }
    if (!g.live[q]) {
      g.a1[q] = zero;
      if (TWO) g.a2[q] = zero;
    }
  }
  // two adjacent rows per 32-bit write: (q = 0, 1) -> rows 2g, 2g+1;  (q = 2, 3) -> rows 2g+16, 2g+17
#pragma unroll
  for (int pr = 0; pr < 2; ++pr) {
    const int row = 2 * (threadIdx.x >> 5) + 16 * pr;
    const float4 x0 = g.a1[2 * pr], x1 = g.a1[2 * pr + 1];
    *reinterpret_cast<unsigned*>(&L.a1[0][c][row]) = pack_bf16(x0.x, x1.x);
    *reinterpret_cast<unsigned*>(&L.a1[1][c][row]) = pack_bf16(x0.y, x1.y);
    *reinterpret_cast<unsigned*>(&L.a1[2][c][row]) = pack_bf16(x0.z, x1.z);
    *reinterpret_cast<unsigned*>(&L.a1[3][c][row]) = pack_bf16(x0.w, x1.w);
    const float4 y0 = g.b1[2 * pr], y1 = g.b1[2 * pr + 1];
    *reinterpret_cast<unsigned*>(&L.b1[0][c][row]) = pack_bf16(y0.x, y1.x);
    if (!NB32) {
      *reinterpret_cast<unsigned*>(&L.b1[1][c][row]) = pack_bf16(y0.y, y1.y);
      *reinterpret_cast<unsigned*>(&L.b1[2][c][row]) = pack_bf16(y0.z, y1.z);
      *reinterpret_cast<unsigned*>(&L.b1[3][c][row]) = pack_bf16(y0.w, y1.w);
    }
    if (TWO) {
      const float4 u0 = g.a2[2 * pr], u1 = g.a2[2 * pr + 1];
      *reinterpret_cast<unsigned*>(&L.a2[0][c][row]) = pack_bf16(u0.x, u1.x);
      *reinterpret_cast<unsigned*>(&L.a2[1][c][row]) = pack_bf16(u0.y, u1.y);
      *reinterpret_cast<unsigned*>(&L.a2[2][c][row]) = pack_bf16(u0.z, u1.z);
      *reinterpret_cast<unsigned*>(&L.a2[3][c][row]) = pack_bf16(u0.w, u1.w);
      const float4 z0 = g.b2[2 * pr], z1 = g.b2[2 * pr + 1];
      *reinterpret_cast<unsigned*>(&L.b2[0][c][row]) = pack_bf16(z0.x, z1.x);
      if (!NB32) {
        *reinterpret_cast<unsigned*>(&L.b2[1][c][row]) = pack_bf16(z0.y, z1.y);
        *reinterpret_cast<unsigned*>(&L.b2[2][c][row]) = pack_bf16(z0.z, z1.z);
        *reinterpret_cast<unsigned*>(&L.b2[3][c][row]) = pack_bf16(z0.w, z1.w);
      }
    }
  }
}
template <int TYPE, bool TWO, bool NB32>
__device__ __forceinline__ void wb_run(const WgProb& P, WbLds* lds, int r_beg, int r_end, int w, int c, int h, f32x16 (&acc)[4]) {
  WbStage g;
  wb_fetch<TYPE, TWO, NB32>(P, r_beg, r_end, r_beg, g);
  wb_finish_commit<TYPE, TWO, NB32>(P, lds[0], g);
  __syncthreads();
  int buf = 0;
  for (int r0 = r_beg; r0 < r_end; r0 += WB_R) {
    const bool more = r0 + WB_R < r_end;
    if (more) wb_fetch<TYPE, TWO, NB32>(P, r0 + WB_R, r_end, r_beg, g);
    __builtin_amdgcn_sched_barrier(0);
    const WbLds& L = lds[buf];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int k0 = 16 * s + 8 * h;             // this lane's 8 rows of the 16-row step
      const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(&L.a1[w][c][k0]);
      bf16x8 a2;
      if (TWO) a2 = *reinterpret_cast<const bf16x8*>(&L.a2[w][c][k0]);
#pragma unroll
      for (int q = 0; q < (NB32 ? 1 : 4); ++q) {
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, *reinterpret_cast<const bf16x8*>(&L.b1[q][c][k0]), acc[q], 0, 0, 0);
        if (TWO)
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, *reinterpret_cast<const bf16x8*>(&L.b2[q][c][k0]), acc[q], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (more) wb_finish_commit<TYPE, TWO, NB32>(P, lds[buf ^ 1], g);
    __syncthreads();
    buf ^= 1;
  }
}

__global__ void __launch_bounds__(256)
wgrad_bf16_kernel(const WgProb* __restrict__ probs, int chunks, float* __restrict__ slabs, int pair_rows) {
  extern __shared__ __attribute__((aligned(16))) char wg_lds_raw[];
  WbLds* lds = reinterpret_cast<WbLds*>(wg_lds_raw);   // [2]
  WgProb P = probs[blockIdx.y];
  if (P.M < 0) P.M = pair_rows;
  if (!P.lda1) P.lda1 = NF;
  if (!P.lda2) P.lda2 = NF;
  if (!P.ldb1) P.ldb1 = NF;
  if (!P.ldb2) P.ldb2 = NF;
  if (!P.ldh) P.ldh = NF;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 31, h = lane >> 5;
  const int M = P.M;
  int per, nch;
  wg_partition(P, M, chunks, WB_R, per, nch);
  if ((int)blockIdx.x >= nch) return;            // (uniform: this problem uses fewer workgroups than the grid has)
  const int r_beg = blockIdx.x * per, r_end = min(M, r_beg + per);
  f32x16 acc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[q][k] = 0.f;
  const bool two = P.A2 != nullptr;
  const bool nb32 = P.b_cols32 != 0;
  if (r_beg < r_end) {
    if (nb32) {
      if (two) wb_run<WG_PLAIN, true, true>(P, lds, r_beg, r_end, w, c, h, acc);
      else wb_run<WG_PLAIN, false, true>(P, lds, r_beg, r_end, w, c, h, acc);
    } else if (P.type == WG_ACT) {
      if (two) wb_run<WG_ACT, true, false>(P, lds, r_beg, r_end, w, c, h, acc);
      else wb_run<WG_ACT, false, false>(P, lds, r_beg, r_end, w, c, h, acc);
    } else if (P.type == WG_TDACT && two) {
      wb_run<WG_TDACT, true, false>(P, lds, r_beg, r_end, w, c, h, acc);
    } else {
      if (two) wb_run<WG_PLAIN, true, false>(P, lds, r_beg, r_end, w, c, h, acc);
      else wb_run<WG_PLAIN, false, false>(P, lds, r_beg, r_end, w, c, h, acc);
    }
  }
  float* slab = slabs + ((size_t)blockIdx.y * chunks + blockIdx.x) * NF * NF;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int row = (k & 3) + 8 * (k >> 2) + 4 * h;
    const int o = 4 * row + w;
    if (nb32)
      slab[(size_t)o * NF + c] = acc[0][k];
    else
      st4(slab + (size_t)o * NF + 4 * c, make_float4(acc[0][k], acc[1][k], acc[2][k], acc[3][k]));
  }
}

// ---------------------------------------------------------------------------------------------
// fp32-GRADE products on the bf16 matrix pipe (the default of fp32 training since round 3).  v_mfma_f32_32x32x2_f32 made the
// batched launch the largest kernel of a large-batch step (1.5 ms of 7.0 at 1024 aspirin conformers, 0.56 of the fp32-MFMA
// peak); the bf16 form above is traffic-bound but carries 8 bits.  Here every fp32 operand is written as THREE bf16 pieces
//     v = h + m + l,   h = bf16(v),  m = bf16(v - h),  l = bf16(v - h - m)          (24 significant bits, bf16's 8 exponent bits:
// no scaling, no range problem) and a product as the six terms of order <= 2^-16:
//     a b ~ ha hb + (ha mb + ma hb) + (ha lb + ma mb + la hb)                       (6 x v_mfma_f32_32x32x16_bf16, fp32 accumulate)
// The dropped terms are below 3 x 2^-24 |a b|: the class of one fp32 rounding.  Six 8-pass MFMAs replace eight 16-pass ones per
// 16 rows (2.7x less pipe time): the kernel becomes bound by its operand traffic.  Same staging as the bf16 kernel (transposed
// images) with three planes per operand and trips of 16 rows: 72 KiB, one buffer, TWO workgroups per CU -- the next trip's rows are
// in registers while this trip's MFMAs run and are committed behind a barrier, and the other workgroup's MFMAs cover that.
// ---------------------------------------------------------------------------------------------
#define WS_R 16                                  // rows per trip
#define WS_P 24                                  // row pitch (bf16 elements) of one (class, column) line: 16 rows + 8 pad --
                                                 // the 16 lanes of a ds_read_b128 phase then start 12 banks apart: conflict-free
struct WsLds {
  unsigned short a1[3][4][32][WS_P], a2[3][4][32][WS_P], b1[3][4][32][WS_P], b2[3][4][32][WS_P];   // 72 KiB: two workgroups per CU
};
struct WsStage {   // rows 2g, 2g+1 of the trip (g = t >> 5), lane column c = t & 31
  float4 a1[2], a2[2], b1[2], b2[2], ha[2];
  bool live[2];
};
template <int TYPE, bool TWO, bool NB32>
__device__ __forceinline__ void ws_fetch(const WgProb& P, int r0, int r_end, int r_safe, WsStage& g) {
  const int c = threadIdx.x & 31;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int r = r0 + 2 * (threadIdx.x >> 5) + q;
    g.live[q] = r < r_end;
    const int rr = g.live[q] ? r : r_safe;
    g.a1[q] = ld4(P.A1 + (size_t)rr * P.lda1 + 4 * c);
    if (NB32) {
      g.b1[q].x = P.B1[(size_t)rr * P.ldb1 + c];
      if (TWO) g.b2[q].x = P.B2[(size_t)rr * P.ldb2 + c];
    } else if (TYPE == WG_ACT) {
      g.b1[q] = ld4(P.hB + (size_t)rr * P.ldh + 4 * c);
      if (TWO) g.b2[q] = ld4(P.dhB + (size_t)rr * P.ldh + 4 * c);
    } else {
      g.b1[q] = ld4(P.B1 + (size_t)rr * P.ldb1 + 4 * c);
      if (TWO) g.b2[q] = ld4(P.B2 + (size_t)rr * P.ldb2 + 4 * c);
    }
    if (TWO) {
      g.a2[q] = ld4(P.A2 + (size_t)rr * P.lda2 + 4 * c);
      if (TYPE == WG_TDACT) g.ha[q] = ld4(P.hA + (size_t)rr * P.ldh + 4 * c);
    }
  }
}
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned (&o)[3]) {   // rows r, r + 1 of one column, three planes
  const __bf16 h0 = (__bf16)x0, h1 = (__bf16)x1;
  const float r0 = x0 - (float)h0, r1 = x1 - (float)h1;
  const __bf16 m0 = (__bf16)r0, m1 = (__bf16)r1;
  const __bf16 l0 = (__bf16)(r0 - (float)m0), l1 = (__bf16)(r1 - (float)m1);
  bf16x2 v;
  v[0] = h0; v[1] = h1; o[0] = __builtin_bit_cast(unsigned, v);
  v[0] = m0; v[1] = m1; o[1] = __builtin_bit_cast(unsigned, v);
  v[0] = l0; v[1] = l1; o[2] = __builtin_bit_cast(unsigned, v);
}
#define WS_PUT(ARR, CLS, X0, X1)                                                      \
  {                                                                                   \
    unsigned o_[3];                                                                   \
    split3_pair(X0, X1, o_);                                                          \
    *reinterpret_cast<unsigned*>(&L.ARR[0][CLS][c][row]) = o_[0];                     \
    *reinterpret_cast<unsigned*>(&L.ARR[1][CLS][c][row]) = o_[1];                     \
    *reinterpret_cast<unsigned*>(&L.ARR[2][CLS][c][row]) = o_[2];                     \
  }
template <int TYPE, bool TWO, bool NB32>
__device__ __forceinline__ void ws_finish_commit(const WgProb& P, WsLds& L, WsStage& g) {
  const int act = P.activation;
  const int c = threadIdx.x & 31;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    if (!NB32 && TYPE == WG_ACT) {
      const float4 hv = g.b1[q], dh = g.b2[q];
      g.b1[q] = make_float4(act_any(hv.x, act), act_any(hv.y, act), act_any(hv.z, act), act_any(hv.w, act));
      if (TWO)
        g.b2[q] = make_float4(dact_any(hv.x, act) * dh.x, dact_any(hv.y, act) * dh.y, dact_any(hv.z, act) * dh.z,
                              dact_any(hv.w, act) * dh.w);
    }
    if (TWO && TYPE == WG_TDACT) {
      const float4 hv = g.ha[q];
      g.a2[q] = make_float4(g.a2[q].x * dact_any(hv.x, act), g.a2[q].y * dact_any(hv.y, act), g.a2[q].z * dact_any(hv.z, act),
                            g.a2[q].w * dact_any(hv.w, act));
    }
    if (!g.live[q]) {
      g.a1[q] = zero;
      if (TWO) g.a2[q] = zero;
    }
  }
  const int row = 2 * (threadIdx.x >> 5);
  const float4 x0 = g.a1[0], x1 = g.a1[1];
  WS_PUT(a1, 0, x0.x, x1.x) WS_PUT(a1, 1, x0.y, x1.y) WS_PUT(a1, 2, x0.z, x1.z) WS_PUT(a1, 3, x0.w, x1.w)
  const float4 y0 = g.b1[0], y1 = g.b1[1];
  WS_PUT(b1, 0, y0.x, y1.x)
  if (!NB32) {
    WS_PUT(b1, 1, y0.y, y1.y) WS_PUT(b1, 2, y0.z, y1.z) WS_PUT(b1, 3, y0.w, y1.w)
  }
  if (TWO) {
    const float4 u0 = g.a2[0], u1 = g.a2[1];
    WS_PUT(a2, 0, u0.x, u1.x) WS_PUT(a2, 1, u0.y, u1.y) WS_PUT(a2, 2, u0.z, u1.z) WS_PUT(a2, 3, u0.w, u1.w)
    const float4 z0 = g.b2[0], z1 = g.b2[1];
    WS_PUT(b2, 0, z0.x, z1.x)
    if (!NB32) {
      WS_PUT(b2, 1, z0.y, z1.y) WS_PUT(b2, 2, z0.z, z1.z) WS_PUT(b2, 3, z0.w, z1.w)
    }
  }
}
// six MFMAs: the product terms of order <= 2^-16, smallest first
#define WS_MMA(ACC, AH, AM, AL, BARR, Q)                                                                          \
  {                                                                                                               \
    const bf16x8 bh_ = *reinterpret_cast<const bf16x8*>(&L.BARR[0][Q][c][k0]);                                    \
    const bf16x8 bm_ = *reinterpret_cast<const bf16x8*>(&L.BARR[1][Q][c][k0]);                                    \
    const bf16x8 bl_ = *reinterpret_cast<const bf16x8*>(&L.BARR[2][Q][c][k0]);                                    \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AL, bh_, ACC, 0, 0, 0);                                         \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AM, bm_, ACC, 0, 0, 0);                                         \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH, bl_, ACC, 0, 0, 0);                                         \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AM, bh_, ACC, 0, 0, 0);                                         \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH, bm_, ACC, 0, 0, 0);                                         \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH, bh_, ACC, 0, 0, 0);                                         \
  }
template <int TYPE, bool TWO, bool NB32>
__device__ __forceinline__ void ws_run(const WgProb& P, WsLds& L, int r_beg, int r_end, int w, int c, int h, f32x16 (&acc)[4]) {
  WsStage g;
  ws_fetch<TYPE, TWO, NB32>(P, r_beg, r_end, r_beg, g);
  ws_finish_commit<TYPE, TWO, NB32>(P, L, g);
  __syncthreads();
  const int k0 = 8 * h;                          // this lane's 8 rows of the 16-row trip
  for (int r0 = r_beg; r0 < r_end; r0 += WS_R) {
    const bool more = r0 + WS_R < r_end;
    if (more) ws_fetch<TYPE, TWO, NB32>(P, r0 + WS_R, r_end, r_beg, g);   // in flight under the MFMAs below
    __builtin_amdgcn_sched_barrier(0);
    const bf16x8 a1h = *reinterpret_cast<const bf16x8*>(&L.a1[0][w][c][k0]);
    const bf16x8 a1m = *reinterpret_cast<const bf16x8*>(&L.a1[1][w][c][k0]);
    const bf16x8 a1l = *reinterpret_cast<const bf16x8*>(&L.a1[2][w][c][k0]);
    bf16x8 a2h, a2m, a2l;
    if (TWO) {
      a2h = *reinterpret_cast<const bf16x8*>(&L.a2[0][w][c][k0]);
      a2m = *reinterpret_cast<const bf16x8*>(&L.a2[1][w][c][k0]);
      a2l = *reinterpret_cast<const bf16x8*>(&L.a2[2][w][c][k0]);
    }
#pragma unroll
    for (int q = 0; q < (NB32 ? 1 : 4); ++q) {
      WS_MMA(acc[q], a1h, a1m, a1l, b1, q)
      if (TWO) WS_MMA(acc[q], a2h, a2m, a2l, b2, q)
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                             // every wave is done reading this trip's images
    if (more) {
      ws_finish_commit<TYPE, TWO, NB32>(P, L, g);
      __syncthreads();
    }
  }
}

__global__ void __launch_bounds__(256, 2)
wgrad_split_kernel(const WgProb* __restrict__ probs, int chunks, float* __restrict__ slabs, int pair_rows) {
  extern __shared__ __attribute__((aligned(16))) char wg_lds_raw[];
  WsLds& L = *reinterpret_cast<WsLds*>(wg_lds_raw);
  WgProb P = probs[blockIdx.y];
  if (P.M < 0) P.M = pair_rows;
  if (!P.lda1) P.lda1 = NF;
  if (!P.lda2) P.lda2 = NF;
  if (!P.ldb1) P.ldb1 = NF;
  if (!P.ldb2) P.ldb2 = NF;
  if (!P.ldh) P.ldh = NF;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 31, h = lane >> 5;
  const int M = P.M;
  int per, nch;
  wg_partition(P, M, chunks, WS_R, per, nch);
  if ((int)blockIdx.x >= nch) return;            // (uniform: this problem uses fewer workgroups than the grid has)
  const int r_beg = blockIdx.x * per, r_end = min(M, r_beg + per);
  f32x16 acc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[q][k] = 0.f;
  const bool two = P.A2 != nullptr;
  const bool nb32 = P.b_cols32 != 0;
  if (r_beg < r_end) {
    if (nb32) {
      if (two) ws_run<WG_PLAIN, true, true>(P, L, r_beg, r_end, w, c, h, acc);
      else ws_run<WG_PLAIN, false, true>(P, L, r_beg, r_end, w, c, h, acc);
    } else if (P.type == WG_ACT) {
      if (two) ws_run<WG_ACT, true, false>(P, L, r_beg, r_end, w, c, h, acc);
      else ws_run<WG_ACT, false, false>(P, L, r_beg, r_end, w, c, h, acc);
    } else if (P.type == WG_TDACT && two) {
      ws_run<WG_TDACT, true, false>(P, L, r_beg, r_end, w, c, h, acc);
    } else {
      if (two) ws_run<WG_PLAIN, true, false>(P, L, r_beg, r_end, w, c, h, acc);
      else ws_run<WG_PLAIN, false, false>(P, L, r_beg, r_end, w, c, h, acc);
    }
  }
  float* slab = slabs + ((size_t)blockIdx.y * chunks + blockIdx.x) * NF * NF;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int row = (k & 3) + 8 * (k >> 2) + 4 * h;
    const int o = 4 * row + w;
    if (nb32)
      slab[(size_t)o * NF + c] = acc[0][k];
    else
      st4(slab + (size_t)o * NF + 4 * c, make_float4(acc[0][k], acc[1][k], acc[2][k], acc[3][k]));
  }
}

__global__ void __launch_bounds__(256)
wgrad_reduce_kernel(const WgProb* __restrict__ probs, int chunks, const float* __restrict__ slabs, int pair_rows, int trip) {
  const WgProb& P = probs[blockIdx.y];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;   // one output element
  if (t >= NF * NF) return;
  const int o = t / NF, i = t % NF;
  const int ncols = P.ncols ? P.ncols : NF;
  if (i >= ncols) return;
  const float* slab = slabs + (size_t)blockIdx.y * chunks * NF * NF + t;
  float s = 0.f;
  int per, nch;
  wg_partition(P, P.M < 0 ? pair_rows : P.M, chunks, trip, per, nch);
  for (int k = 0; k < nch; ++k) s += slab[(size_t)k * NF * NF];
  P.out[(size_t)o * (P.ldo ? P.ldo : ncols) + i] = s;
}

// column sums of [M][128] arrays (bias gradients, dL/d w4): one workgroup per problem, fixed summation order
// Column sums in two passes (fixed order, no atomics): pass 1, grid (chunks, problems), a workgroup sums its slice of rows
// (4 row groups x 128 columns) into part[problem][chunk][128]; pass 2 adds the chunks.
#define CS_CHUNKS 64
__global__ void __launch_bounds__(512)
colsum_partial_kernel(const nnhip_colsum_problem* __restrict__ probs, float* __restrict__ part) {
  __shared__ float sh[4][NF];
  const nnhip_colsum_problem P = probs[blockIdx.y];
  const int col = threadIdx.x & (NF - 1), g = threadIdx.x >> 7;
  const int per = (P.rows + CS_CHUNKS - 1) / CS_CHUNKS;
  const int r0 = blockIdx.x * per, r1 = min(P.rows, r0 + per);
  float s = 0.f;
  for (int r = r0 + g; r < r1; r += 4) s += P.src[(size_t)r * NF + col];
  sh[g][col] = s;
  __syncthreads();
  if (g == 0) part[((size_t)blockIdx.y * CS_CHUNKS + blockIdx.x) * NF + col] = (sh[0][col] + sh[1][col]) + (sh[2][col] + sh[3][col]);
}
__global__ void __launch_bounds__(NF)
colsum_final_kernel(const nnhip_colsum_problem* __restrict__ probs, const float* __restrict__ part) {
  const float* p = part + (size_t)blockIdx.x * CS_CHUNKS * NF + threadIdx.x;
  float s = 0.f;
  for (int k = 0; k < CS_CHUNKS; ++k) s += p[(size_t)k * NF];
  probs[blockIdx.x].out[threadIdx.x] = s;
}

// Per-element sums  out[zz][c] = sum_{i : z_i = zz} x[i][c]  (embedding / scale / shift gradients), two passes: a workgroup
// walks its slice of atoms and accumulates rows into an LDS table [119][width] (thread = column: no conflicts, fixed order);
// pass 2 adds the per-workgroup tables.  width <= 128.
#define SP_CHUNKS 256
#ifndef SP_ATOMS
#define SP_ATOMS 64
#endif
static inline int species_chunks(int n_atoms) {   // ~SP_ATOMS atoms per workgroup, at most SP_CHUNKS
  const int c = (n_atoms + SP_ATOMS - 1) / SP_ATOMS;
  return c < 1 ? 1 : (c > SP_CHUNKS ? SP_CHUNKS : c);
}
__global__ void __launch_bounds__(NF)
species_partial_kernel(const float* __restrict__ x, int ldx, int width, const int64_t* __restrict__ z, int n_atoms,
                       float* __restrict__ part /*[chunks][119][width]*/) {
  extern __shared__ float tab[];   // [119][width]
  const int col = threadIdx.x;
  for (int k = col; k < NNHIP_N_ELEMENTS * width; k += NF) tab[k] = 0.f;
  __syncthreads();
  const int per = (n_atoms + gridDim.x - 1) / gridDim.x;
  const int i0 = blockIdx.x * per, i1 = min(n_atoms, i0 + per);
  if (col < width)
    for (int i = i0; i < i1; ++i) tab[(int)z[i] * width + col] += x[(size_t)i * ldx + col];
  __syncthreads();
  for (int k = col; k < NNHIP_N_ELEMENTS * width; k += NF) part[(size_t)blockIdx.x * NNHIP_N_ELEMENTS * width + k] = tab[k];
}
// out[zz][c] (pitch ldo) = sum over chunks; `cols` columns starting at part column c0.  1024 threads: 8 groups walk the chunks
// 8 apart, the group sums are added in a fixed order.
__global__ void __launch_bounds__(1024)
species_final_kernel(const float* __restrict__ part, int chunks, int width, int c0, int cols, float* __restrict__ out, int ldo) {
  __shared__ float sh[8][NF];
  const int zz = blockIdx.x, c = threadIdx.x & (NF - 1), g = threadIdx.x >> 7;
  float s = 0.f;
  if (c < cols)
    for (int k = g; k < chunks; k += 8) s += part[((size_t)k * NNHIP_N_ELEMENTS + zz) * width + c0 + c];
  sh[g][c] = s;
  __syncthreads();
  if (g == 0 && c < cols) {
#pragma unroll
    for (int q = 1; q < 8; ++q) s += sh[q][c];
    out[(size_t)zz * ldo + c] = s;
  }
}
// out[0] = sum over chunks and elements of column c0 (a plain sum over atoms, e.g. dL/d b4); fixed order
__global__ void __launch_bounds__(1024)
species_total_kernel(const float* __restrict__ part, int chunks, int width, int c0, float* __restrict__ out) {
  __shared__ float sh[1024];
  float s = 0.f;
  for (int k = threadIdx.x; k < chunks * NNHIP_N_ELEMENTS; k += 1024) s += part[(size_t)k * width + c0];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = sh[0];
}

// Small batches (the launch-bound regime): the same sums in ONE launch, no table -- workgroup zz first lists the atoms of its
// element (each of its two waves scans a strided half of z with coalesced loads and compacts its hits by ballot: fixed order),
// then adds their rows; workgroup 119 forms the plain total.
#define SP_DIRECT_MAX_ATOMS 1024
__global__ void __launch_bounds__(NF)
species_direct_kernel(const float* __restrict__ x, int ldx, int width, const int64_t* __restrict__ z, int n_atoms,
                      float* __restrict__ out0, int c0, int cols0, int ldo0, float* __restrict__ out1, int c1, int cols1, int ldo1,
                      float* __restrict__ total, int c_total) {
  const int zz = blockIdx.x, c = threadIdx.x;
  if (zz < NNHIP_N_ELEMENTS) {
    __shared__ unsigned short list[2][SP_DIRECT_MAX_ATOMS];
    __shared__ int cnt[2];
    const int wave = c >> 6, lane = c & 63;
    int n = 0;   // wave-uniform
    for (int base = wave * 64; base < n_atoms; base += NF) {
      const int i = base + lane;
      const bool hit = i < n_atoms && (int)z[i] == zz;
      const unsigned long long m = __ballot(hit);
      if (hit) list[wave][n + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)i;
      n += __popcll(m);
    }
    if (lane == 0) cnt[wave] = n;
    __syncthreads();
    float s = 0.f;
    if (c < width) {
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        const int m = cnt[w];
#pragma unroll 4
        for (int k = 0; k < m; ++k) s += x[(size_t)list[w][k] * ldx + c];
      }
    }
    if (out0 && c >= c0 && c < c0 + cols0) out0[(size_t)zz * ldo0 + (c - c0)] = s;
    if (out1 && c >= c1 && c < c1 + cols1) out1[(size_t)zz * ldo1 + (c - c1)] = s;
  } else if (total) {
    __shared__ float sh[NF];
    float s = 0.f;
    for (int i = c; i < n_atoms; i += NF) s += x[(size_t)i * ldx + c_total];
    sh[c] = s;
    __syncthreads();
    for (int o = NF / 2; o > 0; o >>= 1) {
      if (c < o) sh[c] += sh[c + o];
      __syncthreads();
    }
    if (c == 0) total[0] = sh[0];
  }
}

// out[i][:] = table[z[i]][:]   (node embedding lookup, newtonnet.py:142)
__global__ void __launch_bounds__(256)
embed_rows_kernel(const int64_t* __restrict__ z, const float* __restrict__ table, int n_atoms, float* __restrict__ out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)n_atoms * (NF / 4)) return;
  const size_t i = t / (NF / 4), c = t % (NF / 4);
  reinterpret_cast<float4*>(out)[t] = reinterpret_cast<const float4*>(table + (size_t)z[i] * NF)[c];
}

// ---------------------------------------------------------------------------------------------
// The reference's training objective and its gradient in one launch (newtonnet/train/loss.py:53-103: BaseLoss with
// nn.MSELoss / nn.L1Loss / nn.HuberLoss, mean reduction; scripts/config.yml:45-51):
//   loss = w[0] sum_b l_E(E_b - E*_b) + w[1] sum_{i,k} l_F(F_ik - F*_ik)    w = (w_E / n_E, w_F / n_F) read from DEVICE memory
//   gE = w[0] l_E'(E - E*),  gF = w[1] l_F'(F - F*)                          (data-parallel runs refresh the global counts there)
//   mse: l = d^2; mae: l = |d| (l'(0) = 0 like torch.sign); huber(delta): l = d^2/2 for |d| <= delta, delta (|d| - delta/2) beyond
// One workgroup, fixed summation order (the arrays are a few thousand elements; the step is launch-bound at that size).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float loss_term(float d, int mode, float delta, float& dl) {
  const float a = fabsf(d);
  if (mode == NNHIP_LOSS_MAE) {
    dl = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    return a;
  }
  if (mode == NNHIP_LOSS_HUBER) {
    if (a <= delta) {
      dl = d;
      return 0.5f * d * d;
    }
    dl = d > 0.f ? delta : -delta;
    return delta * (a - 0.5f * delta);
  }
  dl = 2.f * d;
  return d * d;
}
__global__ void __launch_bounds__(1024)
loss_grad_kernel(const float* __restrict__ e, const float* __restrict__ e_lab, int n_e, const float* __restrict__ f,
                 const float* __restrict__ f_lab, int n_f, const float* __restrict__ w, int mode_e, int mode_f, float delta_e,
                 float delta_f, float* __restrict__ loss, float* __restrict__ g_e, float* __restrict__ g_f) {
  __shared__ double sh[1024];
  const float we = w[0], wf = w[1];
  double s = 0.0;
  for (int k = threadIdx.x; k < n_e; k += 1024) {
    float dl;
    const float l = loss_term(e[k] - e_lab[k], mode_e, delta_e, dl);
    g_e[k] = we * dl;
    s += (double)we * l;
  }
  for (int k = threadIdx.x; k < n_f; k += 1024) {
    float dl;
    const float l = loss_term(f[k] - f_lab[k], mode_f, delta_f, dl);
    g_f[k] = wf * dl;
    s += (double)wf * l;
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = (float)sh[0];
}

// ---------------------------------------------------------------------------------------------
// clip_grad_norm_ + Adam on FLAT parameter / gradient / moment buffers (trainer.py:311-313 with torch.optim.Adam defaults:
// no weight decay, no amsgrad), two launches:
//   pass 1: per-workgroup partial sums of g^2 (fixed order); workgroup 0 also advances the device-side step counter
//   pass 2: every workgroup adds the partials in the same order -> total norm, clip = min(1, max_norm / (norm + 1e-6));
//           g' = clip g;  m = b1 m + (1 - b1) g';  v = b2 v + (1 - b2) g'^2;
//           p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// state[0] = step t (float), state[1] = last total norm (for logging).  max_norm <= 0: no clipping.
// hyper (DEVICE, optional): (lr, beta1, beta2, eps, max_norm) read at run time, so a captured HIP graph follows a learning-rate
// schedule; NULL: the by-value arguments.  mask (DEVICE bytes, optional): elements with mask == 0 are frozen parameters
// (requires_grad False): they do not enter the norm and are not updated, like parameters the optimizer was never given.
// ---------------------------------------------------------------------------------------------
#define OPT_BLOCKS 256
__global__ void __launch_bounds__(256)
gradnorm_partial_kernel(const float* __restrict__ g, long n, float* __restrict__ part, float* __restrict__ state,
                        const uint8_t* __restrict__ mask) {
  __shared__ double sh[256];
  const long per = (n + OPT_BLOCKS - 1) / OPT_BLOCKS;
  const long k0 = blockIdx.x * per, k1 = min(n, k0 + per);
  double s = 0.0;
  for (long k = k0 + threadIdx.x; k < k1; k += 256)
    if (!mask || mask[k]) s += (double)g[k] * (double)g[k];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    part[blockIdx.x] = (float)sh[0];
    if (blockIdx.x == 0) state[0] += 1.0f;
  }
}
__global__ void __launch_bounds__(256)
clip_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
                 const float* __restrict__ part, float* __restrict__ state, float lr, float b1, float b2, float eps,
                 float max_norm, const float* __restrict__ hyper, const uint8_t* __restrict__ mask) {
  __shared__ float s_clip, s_c1, s_c2;
  if (hyper) {
    lr = hyper[0];
    b1 = hyper[1];
    b2 = hyper[2];
    eps = hyper[3];
    max_norm = hyper[4];
  }
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int k = 0; k < OPT_BLOCKS; ++k) tot += (double)part[k];
    const float norm = (float)sqrt(tot);
    s_clip = max_norm > 0.f ? fminf(1.0f, max_norm / (norm + 1e-6f)) : 1.0f;
    const float t = state[0];
    s_c1 = lr / (1.0f - powf(b1, t));
    s_c2 = 1.0f / sqrtf(1.0f - powf(b2, t));
    if (blockIdx.x == 0) state[1] = norm;
  }
  __syncthreads();
  const float clip = s_clip, c1 = s_c1, c2 = s_c2;
  const long per = (n + OPT_BLOCKS - 1) / OPT_BLOCKS;
  const long k0 = blockIdx.x * per, k1 = min(n, k0 + per);
  for (long k = k0 + threadIdx.x; k < k1; k += 256) {
    if (mask && !mask[k]) continue;
    const float gk = g[k] * clip;
    const float mk = fmaf(b1, m[k], (1.0f - b1) * gk);
    const float vk = fmaf(b2, v[k], (1.0f - b2) * gk * gk);
    m[k] = mk;
    v[k] = vk;
    p[k] -= c1 * mk / (sqrtf(vk) * c2 + eps);
  }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm in the tangent sweeps (layer_norm=True, newtonnet.py:202-205,228-231; values: edge.hip:layer_norm_fwd/bwd_kernel).
// One wave per atom row, F = 128 features; x_hat and r = 1/sigma are those of the value sweep.
//   tangent forward   s = mean(x_hat dx);  dx_hat = r (dx - mean(dx) - x_hat s);  dr = -r^2 s;  dy = gamma dx_hat
//   tangent of the adjoint g_x = r A,  A = g' - mean(g') - x_hat c,  g' = gamma g_y,  c = mean(g' x_hat):
//                     dA = dg' - mean(dg') - dx_hat c - x_hat mean(dg' x_hat + g' dx_hat);   dg_x = dr A + r dA
//   and the rows whose column sums are the parameter gradients (the epsilon-part of  g_gamma = sum g_y x_hat,  g_beta = sum g_y):
//                     row_w = dg_y x_hat + g_y dx_hat;   row_b = dg_y
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
layer_norm_tan_fwd_kernel(float* __restrict__ da /*in: tangent of the pre-norm row, out: of the normalised row*/,
                          const float* __restrict__ xhat, const float* __restrict__ rstd, const float* __restrict__ gamma,
                          int n_atoms, float* __restrict__ dxhat, float* __restrict__ drstd) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n_atoms) return;
  const int lane = threadIdx.x & 63;
  const float2 dx = ld2(da + (size_t)i * NF + 2 * lane), xh = ld2(xhat + (size_t)i * NF + 2 * lane);
  const float r = rstd[i];
  const float md = wave_sum(dx.x + dx.y) * (1.0f / NF);
  const float sm = wave_sum(fmaf(xh.x, dx.x, xh.y * dx.y)) * (1.0f / NF);
  const float2 dxh = make_float2(r * (dx.x - md - xh.x * sm), r * (dx.y - md - xh.y * sm));
  const float2 g = ld2(gamma + 2 * lane);
  st2(dxhat + (size_t)i * NF + 2 * lane, dxh);
  st2(da + (size_t)i * NF + 2 * lane, make_float2(g.x * dxh.x, g.y * dxh.y));
  if (lane == 0) drstd[i] = -r * r * sm;
}
__global__ void __launch_bounds__(256)
layer_norm_tan_bwd_kernel(const float* __restrict__ gy, float* __restrict__ dga /*in: dg_y, out: dg_x*/,
                          const float* __restrict__ xhat, const float* __restrict__ rstd, const float* __restrict__ dxhat,
                          const float* __restrict__ drstd, const float* __restrict__ gamma, int n_atoms,
                          float* __restrict__ row_w, float* __restrict__ row_b) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n_atoms) return;
  const int lane = threadIdx.x & 63;
  const size_t o = (size_t)i * NF + 2 * lane;
  const float2 g = ld2(gy + o), dg = ld2(dga + o), xh = ld2(xhat + o), dxh = ld2(dxhat + o), w = ld2(gamma + 2 * lane);
  const float r = rstd[i], dr = drstd[i];
  const float2 gp = make_float2(g.x * w.x, g.y * w.y), dgp = make_float2(dg.x * w.x, dg.y * w.y);
  const float m1 = wave_sum(gp.x + gp.y) * (1.0f / NF);
  const float c = wave_sum(fmaf(gp.x, xh.x, gp.y * xh.y)) * (1.0f / NF);
  const float dm1 = wave_sum(dgp.x + dgp.y) * (1.0f / NF);
  const float dc = wave_sum(fmaf(dgp.x, xh.x, dgp.y * xh.y) + fmaf(gp.x, dxh.x, gp.y * dxh.y)) * (1.0f / NF);
  const float2 A = make_float2(gp.x - m1 - xh.x * c, gp.y - m1 - xh.y * c);
  const float2 dA = make_float2(dgp.x - dm1 - dxh.x * c - xh.x * dc, dgp.y - dm1 - dxh.y * c - xh.y * dc);
  st2(row_w + o, make_float2(fmaf(dg.x, xh.x, g.x * dxh.x), fmaf(dg.y, xh.y, g.y * dxh.y)));
  st2(row_b + o, dg);
  st2(dga + o, make_float2(fmaf(dr, A.x, r * dA.x), fmaf(dr, A.y, r * dA.y)));
}
int launch_layer_norm_tan_fwd(float* da, const float* xhat, const float* rstd, const float* gamma, int n_atoms, float* dxhat,
                              float* drstd, hipStream_t s) {
  layer_norm_tan_fwd_kernel<<<cdiv(n_atoms, 4), 256, 0, s>>>(da, xhat, rstd, gamma, n_atoms, dxhat, drstd);
  LAUNCH_CHECK();
  return 0;
}
int launch_layer_norm_tan_bwd(const float* gy, float* dga, const float* xhat, const float* rstd, const float* dxhat,
                              const float* drstd, const float* gamma, int n_atoms, float* row_w, float* row_b, hipStream_t s) {
  layer_norm_tan_bwd_kernel<<<cdiv(n_atoms, 4), 256, 0, s>>>(gy, dga, xhat, rstd, dxhat, drstd, gamma, n_atoms, row_w, row_b);
  LAUNCH_CHECK();
  return 0;
}

// =============================================================================================
// C ABI (include/newtonnet_hip.h, "Training")
// =============================================================================================
#define ARG_CHECK(cond, who)                                   \
  do {                                                         \
    if (!(cond)) {                                             \
      nnhip_set_error("%s: bad arguments (%s)", who, #cond);   \
      return NNHIP_E_INVALID;                                  \
    }                                                          \
  } while (0)
static inline int rows_grid(int n_atoms) { return cdiv(n_atoms, EDGE_ROWS); }

extern "C" int nnhip_embed(const int64_t* z, const float* table, int32_t n_atoms, float* out, void* stream) {
  ARG_CHECK(n_atoms >= 0 && (n_atoms == 0 || (z && table && out)), "nnhip_embed");
  if (n_atoms == 0) return NNHIP_OK;
  embed_rows_kernel<<<cdiv((long)n_atoms * (NF / 4), 256), 256, 0, (hipStream_t)stream>>>(z, table, n_atoms, out);
  LAUNCH_CHECK();
  return NNHIP_OK;
}

extern "C" int nnhip_edge_tangent_geom(const float* v, float sign, const int64_t* edge_index, const float* geo, int32_t n_edges,
                                       float cutoff, float* tgeo, void* stream) {
  ARG_CHECK(n_edges >= 0 && cutoff > 0.f && (n_edges == 0 || (v && edge_index && geo && tgeo)), "nnhip_edge_tangent_geom");
  if (n_edges == 0) return NNHIP_OK;
  edge_tan_geom_kernel<<<cdiv(n_edges, 256), 256, 0, (hipStream_t)stream>>>(v, sign, edge_index, geo, n_edges, 1.0f / cutoff, tgeo);
  LAUNCH_CHECK();
  return NNHIP_OK;
}

extern "C" int nnhip_message_tan_fwd(const float* m, const float* dm, const int32_t* xg, const float* tgeo,
                                     const float* table, const int32_t* row_ptr, const int32_t* col, const int32_t* pid,
                                     const float* da_in, float* dmsg, float* da_mid, int32_t n_atoms, void* stream) {
  ARG_CHECK(n_atoms >= 0 && m && xg && tgeo && table && row_ptr && col && pid && da_mid && (!dm == !da_in),
            "nnhip_message_tan_fwd");
  if (n_atoms == 0) return NNHIP_OK;
  hipStream_t s = (hipStream_t)stream;
  ScopedTimer t0(TC_EDGE, s);
  const int2* x2 = reinterpret_cast<const int2*>(xg);
  if (dm)
    msg_tan_fwd_kernel<true><<<rows_grid(n_atoms), 64 * EDGE_ROWS, 0, s>>>(m, dm, x2, tgeo, table, row_ptr, col, pid, da_in,
                                                                             dmsg, da_mid, n_atoms);
  else
    msg_tan_fwd_kernel<false><<<rows_grid(n_atoms), 64 * EDGE_ROWS, 0, s>>>(m, dm, x2, tgeo, table, row_ptr, col, pid, da_in,
                                                                              dmsg, da_mid, n_atoms);
  LAUNCH_CHECK();
  return NNHIP_OK;
}

extern "C" int nnhip_force_message_tan_fwd(const float* phi1, const float* dphi1, const float* phi2, const float* dphi2,
                                           const float* geo, const float* tgeo, const int32_t* xg, const int32_t* row_ptr,
                                           const int32_t* col, const int32_t* pid, const float* f_in, const float* df_in,
                                           float* df_out, int32_t n_atoms, void* stream) {
  ARG_CHECK(n_atoms >= 0 && phi1 && dphi1 && geo && tgeo && xg && row_ptr && col && pid && df_out &&
                (!f_in == !df_in) && (!f_in || (phi2 && dphi2)), "nnhip_force_message_tan_fwd");
  if (n_atoms == 0) return NNHIP_OK;
  hipStream_t s = (hipStream_t)stream;
  ScopedTimer t0(TC_EDGE, s);
  const int2* x2 = reinterpret_cast<const int2*>(xg);
  if (f_in)
    force_tan_fwd_kernel<true><<<rows_grid(n_atoms), 64 * EDGE_ROWS, 0, s>>>(phi1, dphi1, phi2, dphi2, geo, tgeo, x2, row_ptr,
                                                                               col, pid, f_in, df_in, df_out, n_atoms);
  else
    force_tan_fwd_kernel<false><<<rows_grid(n_atoms), 64 * EDGE_ROWS, 0, s>>>(phi1, dphi1, phi2, dphi2, geo, tgeo, x2, row_ptr,
                                                                                col, pid, f_in, df_in, df_out, n_atoms);
  LAUNCH_CHECK();
  return NNHIP_OK;
}

extern "C" int nnhip_force_message_tan_bwd(const float* gf, const float* dgf, const float* phi2, const float* dphi2,
                                           const float* geo, const float* tgeo, const int32_t* xg, const int32_t* row_ptr,
                                           const int32_t* col, const int32_t* pid, const float* f_in, const float* df_in,
                                           float* dg_h12, float* dg_fin, int32_t n_atoms, void* stream) {
  ARG_CHECK(n_atoms >= 0 && gf && dgf && geo && tgeo && xg && row_ptr && col && pid && dg_h12 && (!f_in == !df_in) &&
                (!f_in || (phi2 && dphi2 && dg_fin)), "nnhip_force_message_tan_bwd");
  if (n_atoms == 0) return NNHIP_OK;
  hipStream_t s = (hipStream_t)stream;
  ScopedTimer t0(TC_EDGE, s);
  const int2* x2 = reinterpret_cast<const int2*>(xg);
  if (f_in)
    force_tan_bwd_kernel<true><<<rows_grid(n_atoms), 64 * EDGE_ROWS, 0, s>>>(gf, dgf, phi2, dphi2, geo, tgeo, x2, row_ptr, col,
                                                                               pid, f_in, df_in, dg_h12, dg_fin, n_atoms);
  else
    force_tan_bwd_kernel<false><<<rows_grid(n_atoms), 64 * EDGE_ROWS, 0, s>>>(gf, dgf, phi2, dphi2, geo, tgeo, x2, row_ptr, col,
                                                                                pid, f_in, df_in, dg_h12, dg_fin, n_atoms);
  LAUNCH_CHECK();
  return NNHIP_OK;
}

extern "C" int nnhip_message_tan_bwd(const float* g_msg, const float* dg_msg, const float* ga, const float* dga,
                                     const float* m, const float* dm, const int32_t* xg, const float* tgeo,
                                     const float* table, const int32_t* row_ptr, const int32_t* col, const int32_t* pid,
                                     float* dg_m, float* g_eps, float* dg_eps, int32_t n_atoms, void* stream) {
  ARG_CHECK(n_atoms >= 0 && g_msg && dg_msg && ga && dga && m && xg && tgeo && table && row_ptr && col && pid && dg_m &&
                g_eps && dg_eps, "nnhip_message_tan_bwd");
  if (n_atoms == 0) return NNHIP_OK;
  hipStream_t s = (hipStream_t)stream;
  ScopedTimer t0(TC_EDGE, s);
  const int2* x2 = reinterpret_cast<const int2*>(xg);
  if (dm)
    msg_tan_bwd_kernel<true><<<rows_grid(n_atoms), 64 * EDGE_ROWS, 0, s>>>(g_msg, dg_msg, ga, dga, m, dm, x2, tgeo, table,
                                                                             row_ptr, col, pid, dg_m, g_eps, dg_eps, n_atoms);
  else
    msg_tan_bwd_kernel<false><<<rows_grid(n_atoms), 64 * EDGE_ROWS, 0, s>>>(g_msg, dg_msg, ga, dga, m, dm, x2, tgeo, table,
                                                                              row_ptr, col, pid, dg_m, g_eps, dg_eps, n_atoms);
  LAUNCH_CHECK();
  return NNHIP_OK;
}

extern "C" int nnhip_update_tan_fwd(const float* da_mid, const float* f, const float* df, const float* q, const float* dq,
                                    int32_t n_atoms, float* da_out, void* stream) {
  ARG_CHECK(n_atoms >= 0 && da_mid && f && df && q && dq && da_out, "nnhip_update_tan_fwd");
  if (n_atoms == 0) return NNHIP_OK;
  update_tan_fwd_kernel<<<cdiv((long)n_atoms * (NF / 4), 256), 256, 0, (hipStream_t)stream>>>(da_mid, f, df, q, dq, n_atoms,
                                                                                               da_out);
  LAUNCH_CHECK();
  return NNHIP_OK;
}

extern "C" int nnhip_update_tan_bwd(const float* ga, const float* dga, const float* f, const float* df, const float* q,
                                    const float* dq, const float* dgf_in, int32_t n_atoms, float* gq, float* dgq,
                                    float* dgf, void* stream) {
  ARG_CHECK(n_atoms >= 0 && ga && dga && f && df && q && dq && gq && dgq && dgf, "nnhip_update_tan_bwd");
  if (n_atoms == 0) return NNHIP_OK;
  update_tan_bwd_kernel<<<cdiv((long)n_atoms * (NF / 4), 256), 256, 0, (hipStream_t)stream>>>(ga, dga, f, df, q, dq, dgf_in,
                                                                                               n_atoms, gq, dgq, dgf);
  LAUNCH_CHECK();
  return NNHIP_OK;
}

extern "C" int nnhip_head_seed_tan(const float* e2, const float* de2, const float* w4, const float* b4, const float* scale,
                                   const int64_t* z, const int64_t* batch, const float* g_energy, int32_t n_atoms,
                                   int32_t activation, float* dg_e2, float* w4row, float* scal, void* stream) {
  ARG_CHECK(n_atoms >= 0 && e2 && de2 && w4 && b4 && z && batch && g_energy && dg_e2 && w4row && scal &&
                activation >= NNHIP_ACT_SILU && activation <= NNHIP_ACT_SSP, "nnhip_head_seed_tan");
  if (n_atoms == 0) return NNHIP_OK;
  head_seed_tan_kernel<<<cdiv(n_atoms, 4), 256, 0, (hipStream_t)stream>>>(e2, de2, w4, b4, scale, z, batch, g_energy, n_atoms,
                                                                           activation, dg_e2, w4row, scal);
  LAUNCH_CHECK();
  return NNHIP_OK;
}

extern "C" int nnhip_pair_rbf(const float* rbf, const float* drbf, const float* tgeo, const int64_t* edge_index,
                              const int32_t* pid, int32_t n_edges, int32_t n_basis, float* rb, void* stream) {
  ARG_CHECK(n_edges >= 0 && n_basis >= 1 && n_basis <= NNHIP_MAX_NB && (n_edges == 0 || (rbf && drbf && tgeo && edge_index &&
                pid && rb)), "nnhip_pair_rbf");
  if (n_edges == 0) return NNHIP_OK;
  pair_rbf_kernel<<<cdiv((long)n_edges * 32, 256), 256, 0, (hipStream_t)stream>>>(rbf, drbf, tgeo, edge_index, pid, n_edges,
                                                                                   n_basis, rb);
  LAUNCH_CHECK();
  return NNHIP_OK;
}

extern "C" size_t nnhip_species_scratch_bytes(int32_t width) {
  return width >= 1 && width <= NF ? (size_t)SP_CHUNKS * NNHIP_N_ELEMENTS * width * sizeof(float) : 0;
}
// Per-element sums of x[N][ldx] (first `width` columns): up to three outputs, each a column range of the element table:
//   out_k[zz][0:cols_k] (pitch ldo_k) = sum_{i: z_i = zz} x[i][c0_k : c0_k + cols_k];   total (may be NULL) = sum_i x[i][c_total]
extern "C" int nnhip_species_sum(const float* x, int32_t ldx, int32_t width, const int64_t* z, int32_t n_atoms, float* scratch,
                                 float* out0, int32_t c0, int32_t cols0, int32_t ldo0, float* out1, int32_t c1, int32_t cols1,
                                 int32_t ldo1, float* total, int32_t c_total, void* stream) {
  ARG_CHECK(n_atoms >= 0 && x && z && scratch && width >= 1 && width <= NF && ldx >= width && (!out0 || (c0 >= 0 && c0 + cols0 <= width)) &&
                (!out1 || (c1 >= 0 && c1 + cols1 <= width)) && (!total || (c_total >= 0 && c_total < width)), "nnhip_species_sum");
  hipStream_t s = (hipStream_t)stream;
  if (n_atoms <= SP_DIRECT_MAX_ATOMS) {
    species_direct_kernel<<<NNHIP_N_ELEMENTS + (total ? 1 : 0), NF, 0, s>>>(x, ldx, width, z, n_atoms, out0, c0, cols0, ldo0, out1, c1,
                                                                          cols1, ldo1, total, c_total);
    LAUNCH_CHECK();
    return NNHIP_OK;
  }
  const int chunks = species_chunks(n_atoms);
  species_partial_kernel<<<chunks, NF, NNHIP_N_ELEMENTS * width * sizeof(float), s>>>(x, ldx, width, z, n_atoms, scratch);
  LAUNCH_CHECK();
  if (out0) {
    species_final_kernel<<<NNHIP_N_ELEMENTS, 1024, 0, s>>>(scratch, chunks, width, c0, cols0, out0, ldo0);
    LAUNCH_CHECK();
  }
  if (out1) {
    species_final_kernel<<<NNHIP_N_ELEMENTS, 1024, 0, s>>>(scratch, chunks, width, c1, cols1, out1, ldo1);
    LAUNCH_CHECK();
  }
  if (total) {
    species_total_kernel<<<1, 1024, 0, s>>>(scratch, chunks, width, c_total, total);
    LAUNCH_CHECK();
  }
  return NNHIP_OK;
}

// A batch of weight-gradient products in two launches (accumulate + reduce).  The problem table is DEVICE memory (the caller
// uploads it once per workspace: nothing in the launch path touches the host, so the step captures into a HIP graph).
extern "C" size_t nnhip_wgrad_slab_bytes(int32_t n_problems, int32_t chunks) {
  if (n_problems < 0 || chunks < 1) return 0;
  return (size_t)n_problems * chunks * NF * NF * sizeof(float);
}
extern "C" int nnhip_wgrad_batch(const nnhip_wgrad_problem* probs_dev, int32_t n_problems, int32_t chunks, float* slabs,
                                 int32_t bf16_operands, int32_t pair_rows, void* stream) {
  ARG_CHECK(n_problems >= 0 && chunks >= 1 && pair_rows >= 0 && (n_problems == 0 || (probs_dev && slabs)), "nnhip_wgrad_batch");
  if (n_problems == 0) return NNHIP_OK;
  hipStream_t s = (hipStream_t)stream;
  ScopedTimer t0(TC_LIN, s);
  static const hipError_t attr_rc = hipFuncSetAttribute((const void*)wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                        2 * sizeof(WgLds));
  HIP_TRY(attr_rc);
  static const hipError_t attr_rc2 = hipFuncSetAttribute((const void*)wgrad_bf16_kernel,
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, 2 * sizeof(WbLds));
  HIP_TRY(attr_rc2);
  static const hipError_t attr_rc3 = hipFuncSetAttribute((const void*)wgrad_split_kernel,
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, sizeof(WsLds));
  HIP_TRY(attr_rc3);
  {
    ScopedTimer t1(TC_WGRAD, s);
    if (bf16_operands == 1)
      wgrad_bf16_kernel<<<dim3(chunks, n_problems), 256, 2 * sizeof(WbLds), s>>>(probs_dev, chunks, slabs, pair_rows);
    else if (bf16_operands == 2)
      wgrad_split_kernel<<<dim3(chunks, n_problems), 256, sizeof(WsLds), s>>>(probs_dev, chunks, slabs, pair_rows);
    else
      wgrad_kernel<<<dim3(chunks, n_problems), WG_THREADS, 2 * sizeof(WgLds), s>>>(probs_dev, chunks, slabs, pair_rows);
  }
  LAUNCH_CHECK();
  wgrad_reduce_kernel<<<dim3(NF * NF / 256, n_problems), 256, 0, s>>>(probs_dev, chunks, slabs, pair_rows,
                                                                         bf16_operands == 1 ? WB_R : (bf16_operands == 2 ? WS_R : WG_R));
  LAUNCH_CHECK();
  return NNHIP_OK;
}

extern "C" size_t nnhip_colsum_scratch_bytes(int32_t n) { return n > 0 ? (size_t)n * CS_CHUNKS * NF * sizeof(float) : 0; }
extern "C" int nnhip_colsum_batch(const nnhip_colsum_problem* probs_dev, int32_t n, float* scratch, void* stream) {
  ARG_CHECK(n >= 0 && (n == 0 || (probs_dev && scratch)), "nnhip_colsum_batch");
  if (n == 0) return NNHIP_OK;
  colsum_partial_kernel<<<dim3(CS_CHUNKS, n), 512, 0, (hipStream_t)stream>>>(probs_dev, scratch);
  LAUNCH_CHECK();
  colsum_final_kernel<<<n, NF, 0, (hipStream_t)stream>>>(probs_dev, scratch);
  LAUNCH_CHECK();
  return NNHIP_OK;
}

extern "C" int nnhip_loss_grad(const float* energy, const float* energy_label, int32_t n_energy, const float* forces,
                               const float* force_label, int32_t n_force, const float* weights_dev, int32_t mode_energy,
                               int32_t mode_force, float delta_energy, float delta_force, float* loss, float* g_energy,
                               float* g_forces, void* stream) {
  ARG_CHECK(n_energy >= 0 && n_force >= 0 && energy && energy_label && forces && force_label && weights_dev && loss && g_energy &&
                g_forces && mode_energy >= 0 && mode_energy <= NNHIP_LOSS_HUBER && mode_force >= 0 &&
                mode_force <= NNHIP_LOSS_HUBER && delta_energy > 0.f && delta_force > 0.f, "nnhip_loss_grad");
  loss_grad_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(energy, energy_label, n_energy, forces, force_label, n_force, weights_dev,
                                                        mode_energy, mode_force, delta_energy, delta_force, loss, g_energy,
                                                        g_forces);
  LAUNCH_CHECK();
  return NNHIP_OK;
}
extern "C" int nnhip_mse_loss_grad(const float* energy, const float* energy_label, int32_t n_energy, const float* forces,
                                   const float* force_label, int32_t n_force, const float* weights_dev, float* loss,
                                   float* g_energy, float* g_forces, void* stream) {
  return nnhip_loss_grad(energy, energy_label, n_energy, forces, force_label, n_force, weights_dev, NNHIP_LOSS_MSE, NNHIP_LOSS_MSE,
                         1.f, 1.f, loss, g_energy, g_forces, stream);
}

extern "C" size_t nnhip_clip_adam_scratch_bytes(void) { return OPT_BLOCKS * sizeof(float); }
static int clip_adam_launch(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float* scratch,
                            float* state, float lr, float beta1, float beta2, float eps, float max_norm, const float* hyper,
                            const uint8_t* mask, hipStream_t s) {
  if (n == 0) return NNHIP_OK;
  gradnorm_partial_kernel<<<OPT_BLOCKS, 256, 0, s>>>(grads, (long)n, scratch, state, mask);
  LAUNCH_CHECK();
  clip_adam_kernel<<<OPT_BLOCKS, 256, 0, s>>>(params, grads, exp_avg, exp_avg_sq, (long)n, scratch, state, lr, beta1, beta2, eps,
                                              max_norm, hyper, mask);
  LAUNCH_CHECK();
  return NNHIP_OK;
}
extern "C" int nnhip_clip_adam(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float* scratch,
                               float* state, float lr, float beta1, float beta2, float eps, float max_norm, void* stream) {
  ARG_CHECK(n >= 0 && params && grads && exp_avg && exp_avg_sq && scratch && state && lr >= 0.f && beta1 >= 0.f && beta1 < 1.f &&
                beta2 >= 0.f && beta2 < 1.f, "nnhip_clip_adam");
  return clip_adam_launch(params, grads, exp_avg, exp_avg_sq, n, scratch, state, lr, beta1, beta2, eps, max_norm, nullptr, nullptr,
                          (hipStream_t)stream);
}
extern "C" int nnhip_clip_adam_dev(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                                   float* scratch, float* state, const float* hyper_dev, const uint8_t* mask_dev, void* stream) {
  ARG_CHECK(n >= 0 && params && grads && exp_avg && exp_avg_sq && scratch && state && hyper_dev, "nnhip_clip_adam_dev");
  return clip_adam_launch(params, grads, exp_avg, exp_avg_sq, n, scratch, state, 0.f, 0.f, 0.f, 0.f, 0.f, hyper_dev, mask_dev,
                          (hipStream_t)stream);
}
