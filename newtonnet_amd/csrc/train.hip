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
edge_tan_geom_kernel(const float* __restrict__ v, const int64_t* __restrict__ edge_index, const float* __restrict__ geo,
                     int n_edges, float inv_rc, float* __restrict__ tgeo) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_edges) return;
  const long i = edge_index[e], j = edge_index[(long)n_edges + e];
  const float4 g = reinterpret_cast<const float4*>(geo)[e];
  const float dx = v[3 * i] - v[3 * j], dy = v[3 * i + 1] - v[3 * j + 1], dz = v[3 * i + 2] - v[3 * j + 2];
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

struct WgOps {   // operands of one pair of rows for this lane
  float a1, a2;
  float4 b1, b2;
};
__device__ __forceinline__ WgOps wg_load(const WgProb& P, int rr, bool live, int c, int w, bool two) {
  WgOps o;
  o.a2 = 0.f;
  o.b2 = make_float4(0.f, 0.f, 0.f, 0.f);
  const int act = P.activation;
  o.a1 = comp(ld4(P.A1 + (size_t)rr * P.lda1 + 4 * c), w);
  if (P.b_cols32) {
    o.b1 = make_float4(P.B1[(size_t)rr * P.ldb1 + c], 0.f, 0.f, 0.f);
    if (two) o.b2 = make_float4(P.B2[(size_t)rr * P.ldb2 + c], 0.f, 0.f, 0.f);
  } else if (P.type == WG_ACT) {
    const float4 hv = ld4(P.hB + (size_t)rr * P.ldh + 4 * c);
    o.b1 = make_float4(act_any(hv.x, act), act_any(hv.y, act), act_any(hv.z, act), act_any(hv.w, act));
    if (two) {
      const float4 dh = ld4(P.dhB + (size_t)rr * P.ldh + 4 * c);
      o.b2 = make_float4(dact_any(hv.x, act) * dh.x, dact_any(hv.y, act) * dh.y, dact_any(hv.z, act) * dh.z,
                         dact_any(hv.w, act) * dh.w);
    }
  } else {
    o.b1 = ld4(P.B1 + (size_t)rr * P.ldb1 + 4 * c);
    if (two) o.b2 = ld4(P.B2 + (size_t)rr * P.ldb2 + 4 * c);
  }
  if (two) {
    o.a2 = comp(ld4(P.A2 + (size_t)rr * P.lda2 + 4 * c), w);
    if (P.type == WG_TDACT) o.a2 *= dact_any(comp(ld4(P.hA + (size_t)rr * P.ldh + 4 * c), w), act);
  }
  if (!live) {
    o.a1 = 0.f;
    o.a2 = 0.f;
  }
  return o;
}

__global__ void __launch_bounds__(256)
wgrad_kernel(const WgProb* __restrict__ probs, int chunks, float* __restrict__ slabs) {
  WgProb P = probs[blockIdx.y];
  if (!P.lda1) P.lda1 = NF;
  if (!P.lda2) P.lda2 = NF;
  if (!P.ldb1) P.ldb1 = NF;
  if (!P.ldb2) P.ldb2 = NF;
  if (!P.ldh) P.ldh = NF;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // output-feature class of this wave
  const int c = lane & 31, h = lane >> 5;
  const int M = P.M;
  // rows of this workgroup: whole pairs of rows, contiguous
  const int per = ((M + 2 * chunks - 1) / (2 * chunks)) * 2;
  const int r_beg = blockIdx.x * per, r_end = min(M, r_beg + per);
  f32x16 acc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[q][k] = 0.f;
  const bool two = P.A2 != nullptr;
  const bool nb32 = P.b_cols32 != 0;
  // four pairs of rows per trip: all operands are requested before the first MFMA
  for (int r0 = r_beg; r0 < r_end; r0 += 8) {
    WgOps o[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = r0 + 2 * u + h;
      const bool live = r < r_end;
      o[u] = wg_load(P, live ? r : r_beg, live, c, w, two);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(o[u].a1, o[u].b1.x, acc[0], 0, 0, 0);
      if (!nb32) {
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(o[u].a1, o[u].b1.y, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(o[u].a1, o[u].b1.z, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(o[u].a1, o[u].b1.w, acc[3], 0, 0, 0);
      }
      if (two) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(o[u].a2, o[u].b2.x, acc[0], 0, 0, 0);
        if (!nb32) {
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(o[u].a2, o[u].b2.y, acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(o[u].a2, o[u].b2.z, acc[2], 0, 0, 0);
          acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(o[u].a2, o[u].b2.w, acc[3], 0, 0, 0);
        }
      }
    }
  }
  // D[row][col]: row = (k & 3) + 8 (k >> 2) + 4 h -> output feature 4 row + w; col = c -> input feature 4 c + q
  float* slab = slabs + ((size_t)blockIdx.y * chunks + blockIdx.x) * NF * NF;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int row = (k & 3) + 8 * (k >> 2) + 4 * h;
    const int o = 4 * row + w;
    if (nb32)
      slab[(size_t)o * NF + c] = acc[0][k];     // input "feature" c = basis index
    else
      st4(slab + (size_t)o * NF + 4 * c, make_float4(acc[0][k], acc[1][k], acc[2][k], acc[3][k]));
  }
}

__global__ void __launch_bounds__(256)
wgrad_reduce_kernel(const WgProb* __restrict__ probs, int chunks, const float* __restrict__ slabs) {
  const WgProb& P = probs[blockIdx.y];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;   // one output element
  if (t >= NF * NF) return;
  const int o = t / NF, i = t % NF;
  const int ncols = P.ncols ? P.ncols : NF;
  if (i >= ncols) return;
  const float* slab = slabs + (size_t)blockIdx.y * chunks * NF * NF + t;
  float s = 0.f;
  for (int k = 0; k < chunks; ++k) s += slab[(size_t)k * NF * NF];
  P.out[(size_t)o * (P.ldo ? P.ldo : ncols) + i] = s;
}

// column sums of [M][128] arrays (bias gradients, dL/d w4): one workgroup per problem, fixed summation order
__global__ void __launch_bounds__(1024) colsum_kernel(const nnhip_colsum_problem* __restrict__ probs) {
  __shared__ float part[8][NF];
  const nnhip_colsum_problem P = probs[blockIdx.x];
  const int col = threadIdx.x & (NF - 1), g = threadIdx.x >> 7;   // 8 row groups
  float s = 0.f;
  for (int r = g; r < P.rows; r += 8) s += P.src[(size_t)r * NF + col];
  part[g][col] = s;
  __syncthreads();
  if (g == 0) {
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) tot += part[q][col];
    P.out[col] = tot;
  }
}

// per-element sums:  out[zz][c] = sum_{i : z_i = zz} x[i][c]   (embedding / scale / shift gradients); width <= 128 floats per row.
// One workgroup per element: every thread scans z (N is small next to the edge work) -- deterministic, no atomics.
// z == NULL: a single bin (plain column sums of a narrow array).
__global__ void __launch_bounds__(256)
species_sum_kernel(const float* __restrict__ x, int ldx, int width, const int64_t* __restrict__ z, int n_atoms,
                   float* __restrict__ out, int ldo) {
  __shared__ float part[2][NF];
  const int zz = blockIdx.x;
  const int col = threadIdx.x & (NF - 1), g = threadIdx.x >> 7;   // 2 row groups
  float s = 0.f;
  if (col < width)
    for (int i = g; i < n_atoms; i += 2)
      if (!z || z[i] == zz) s += x[(size_t)i * ldx + col];
  part[g][col] = s;
  __syncthreads();
  if (g == 0 && col < width) out[(size_t)zz * ldo + col] = part[0][col] + part[1][col];
}

// out[i][:] = table[z[i]][:]   (node embedding lookup, newtonnet.py:142)
__global__ void __launch_bounds__(256)
embed_rows_kernel(const int64_t* __restrict__ z, const float* __restrict__ table, int n_atoms, float* __restrict__ out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)n_atoms * (NF / 4)) return;
  const size_t i = t / (NF / 4), c = t % (NF / 4);
  reinterpret_cast<float4*>(out)[t] = reinterpret_cast<const float4*>(table + (size_t)z[i] * NF)[c];
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

extern "C" int nnhip_edge_tangent_geom(const float* v, const int64_t* edge_index, const float* geo, int32_t n_edges,
                                       float cutoff, float* tgeo, void* stream) {
  ARG_CHECK(n_edges >= 0 && cutoff > 0.f && (n_edges == 0 || (v && edge_index && geo && tgeo)), "nnhip_edge_tangent_geom");
  if (n_edges == 0) return NNHIP_OK;
  edge_tan_geom_kernel<<<cdiv(n_edges, 256), 256, 0, (hipStream_t)stream>>>(v, edge_index, geo, n_edges, 1.0f / cutoff, tgeo);
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

extern "C" int nnhip_species_sum(const float* x, int32_t ldx, int32_t width, const int64_t* z, int32_t n_atoms, float* out,
                                 int32_t ldo, void* stream) {
  ARG_CHECK(n_atoms >= 0 && x && out && width >= 1 && width <= NF && ldx >= width && ldo >= width, "nnhip_species_sum");
  species_sum_kernel<<<z ? NNHIP_N_ELEMENTS : 1, 256, 0, (hipStream_t)stream>>>(x, ldx, width, z, n_atoms, out, ldo);
  LAUNCH_CHECK();
  return NNHIP_OK;
}

// A batch of weight-gradient products in two launches (accumulate + reduce).  The problem table is DEVICE memory (the caller
// uploads it once per workspace: nothing in the launch path touches the host, so the step captures into a HIP graph).
extern "C" size_t nnhip_wgrad_slab_bytes(int32_t n_problems, int32_t chunks) {
  if (n_problems < 0 || chunks < 1) return 0;
  return (size_t)n_problems * chunks * NF * NF * sizeof(float);
}
extern "C" int nnhip_wgrad_batch(const nnhip_wgrad_problem* probs_dev, int32_t n_problems, int32_t chunks, float* slabs,
                                 void* stream) {
  ARG_CHECK(n_problems >= 0 && chunks >= 1 && (n_problems == 0 || (probs_dev && slabs)), "nnhip_wgrad_batch");
  if (n_problems == 0) return NNHIP_OK;
  hipStream_t s = (hipStream_t)stream;
  ScopedTimer t0(TC_LIN, s);
  wgrad_kernel<<<dim3(chunks, n_problems), 256, 0, s>>>(probs_dev, chunks, slabs);
  LAUNCH_CHECK();
  wgrad_reduce_kernel<<<dim3(NF * NF / 256, n_problems), 256, 0, s>>>(probs_dev, chunks, slabs);
  LAUNCH_CHECK();
  return NNHIP_OK;
}

extern "C" int nnhip_colsum_batch(const nnhip_colsum_problem* probs_dev, int32_t n, void* stream) {
  ARG_CHECK(n >= 0 && (n == 0 || probs_dev), "nnhip_colsum_batch");
  if (n == 0) return NNHIP_OK;
  colsum_kernel<<<n, 1024, 0, (hipStream_t)stream>>>(probs_dev);
  LAUNCH_CHECK();
  return NNHIP_OK;
}
