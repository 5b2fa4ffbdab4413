// Single-launch energy + force step for SMALL systems (the MD loop of MLAseCalculator.calculate, utils/ase_interface.py:52-81,
// caller scripts/simulate.py:21-30: one 21-atom molecule per call).
//
// At this size the multi-kernel path of pipeline.hip is bound by the latency of ~31 DEPENDENT launches (5-18 us each, 280 us
// in total for a 21-atom step; replaying them from a HIP graph changes nothing, the dependency latency stays).  Here the same
// phases -- embedding, per layer: messages + invariant sum, the two edge MLPs, equivariant messages + sum, equiv_update +
// energy update + the next message_nodepart; energy head; and the whole analytic reverse sweep down to the forces
// (newtonnet/models/newtonnet.py:139-161,207-231, output.py:66-73,90-100) -- run inside ONE workgroup of 16 waves, separated by
// workgroup barriers (~0.1 us) instead of kernel boundaries.  The intermediates are the workspace arrays of pipeline.hip (a few
// hundred KB: they never leave L2), the weights are the prepared split-f16 images read straight from L2 as MFMA A operands.
//
// Every dense product goes through ONE routine, wg_gemm: 32-row x 32-column units dealt to the 16 waves, fp32 products formed
// from split-f16 pieces exactly as in node128s.hip / mlp128s.hip (rows scaled by their own maximum, weights per matrix, three
// v_mfma_f32_32x32x16_f16 per 16 k-values, fp32 accumulation).  The edge phases are one wave per receiver row with the same
// pair-once layout, ownership rule and formulas as edge.hip.  Deterministic: fixed summation orders, no atomics.
//
// STATUS (round 3, measured): correct (parity tests against the oracle and against the multi-kernel path) but SLOWER than the
// launches it replaces -- 905 us vs 278 us for the 21-atom aspirin step (tools/bench_latency.py) -- so pipeline.hip selects it
// only under NNHIP_SMALL_STEP=1.  Why: one CU runs ~50 barrier-separated phases, and inside a phase nothing hides the L2 latency
// (~1 us per dependent access): a row phase is two rounds of ~14 dependent edge iterations, a dense phase two to three rounds of
// GEMM units whose operands come from L2.  The matrix work itself (~570 units x 24 MFMAs) is ~50 us on one CU; reaching it
// needs (a) per-tile fusion of consecutive GEMMs with register hand-over as in mlp128s.hip, (b) node-level state in LDS,
// (c) edge loops with their indices preloaded and two rows per wave -- DESIGN.md section 7 has the estimate (150 us at best,
// i.e. not the 2x over the multi-kernel path that would justify it).
//
// Scope: SiLU models without LayerNorm, at most SMALL_MAX_ATOMS atoms and SMALL_MAX_EDGES directed edges in the whole batch,
// energy + forces (no virial); everything else takes the multi-kernel path.  Throughput is not the point of this kernel.
#include <string.h>

#include "common.h"

#include "edge_common.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define SM_WAVES 16
#define SM_THREADS (64 * SM_WAVES)
#define SM_WIMG_PLANE (NF * NF * 2)

enum { SPRO_NONE = 0, SPRO_ACT = 1 };                                   // prologue on X
enum { SEPI_STORE = 0, SEPI_BIAS = 1, SEPI_DACT = 2, SEPI_ACC = 3, SEPI_GF = 4 };

__device__ __forceinline__ void sm_pow2_scale(float m, float& S, float& inv) {
  const int e = (int)((__float_as_uint(m) >> 23) & 0xffu);
  const bool ok = e >= 40 && e < 255;
  S = ok ? __uint_as_float((unsigned)(268 - e) << 23) : 1.0f;
  inv = ok ? __uint_as_float((unsigned)(e - 14) << 23) : 1.0f;
}

// Y[r][0:128] = epi( pro(X[r][0:128]) . W^T ) for r in [0, M); W = prepared split-f16 image (node128s.hip:weight_image_kernel).
//   SPRO_ACT : X <- silu(X)
//   SEPI_BIAS: + bias[col];   SEPI_DACT: * silu'(H[r][col]) (H row pitch 128; Y may alias H);   SEPI_ACC: Y += ...;
//   SEPI_GF  : Y = acc + G[r][col] + ga[r / 3][col] * q[r][col]   (update adjoint; G may be NULL)
// Called by ALL waves of the workgroup; no barrier inside (the caller separates phases).
template <int PRO, int EPI>
__device__ __forceinline__ void wg_gemm(const float* X, int ldx, const char* img, float* Y, int ldy, int M, const float* aux0,
                                        const float* aux1, const float* aux2) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int n_units = ((M + 31) >> 5) * 4;
  const float inv_w = *reinterpret_cast<const float*>(img + 2 * SM_WIMG_PLANE);
  for (int unit = wave; unit < n_units; unit += SM_WAVES) {
    const int tile = unit >> 2, nb = unit & 3;
    const int row = (tile << 5) + r;
    const int rc = min(row, M - 1);
    // this lane's half of the row: features 8 t + 4 h + {0..3}.  Two passes over the row (the second hits L1): the row maximum
    // first, then split operand by operand -- 16 waves leave 128 registers per lane, not enough to hold the row
    const float4* xp = reinterpret_cast<const float4*>(X + (size_t)rc * ldx + 4 * h);
    float m = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      float4 xv = xp[2 * t];
      if (PRO == SPRO_ACT) xv = make_float4(silu_f(xv.x), silu_f(xv.y), silu_f(xv.z), silu_f(xv.w));
      m = fmaxf(fmaxf(fmaxf(m, fabsf(xv.x)), fmaxf(fabsf(xv.y), fabsf(xv.z))), fabsf(xv.w));
    }
    m = fmaxf(m, __shfl_xor(m, 32));
    float S, inv;
    sm_pow2_scale(m, S, inv);
    const float scale_out = inv * inv_w;
    f32x16 acc;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
    const char* wp = img + ((size_t)(nb * 8 * 64 + h * 32 + r) << 4);   // fragment order (node128s.hip:weight_image_kernel)
#pragma unroll
    for (int T = 0; T < 8; ++T) {
      const h8 ah = *reinterpret_cast<const h8*>(wp + 1024 * T);
      const h8 al = *reinterpret_cast<const h8*>(wp + SM_WIMG_PLANE + 1024 * T);
      float4 x0 = xp[4 * T], x1 = xp[4 * T + 2];
      if (PRO == SPRO_ACT) {
        x0 = make_float4(silu_f(x0.x), silu_f(x0.y), silu_f(x0.z), silu_f(x0.w));
        x1 = make_float4(silu_f(x1.x), silu_f(x1.y), silu_f(x1.z), silu_f(x1.w));
      }
      const float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
      h8 bh, bl;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float sv = v[j] * S;
        const _Float16 a = (_Float16)sv;
        bh[j] = a;
        bl[j] = (_Float16)(sv - (float)a);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
    }
    if (row < M) {
      // acc[4 q + c] = output feature nb * 32 + 8 q + 4 h + c of this lane's row
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const int c0 = nb * 32 + 8 * qq + 4 * h;
        float4 v = make_float4(acc[4 * qq] * scale_out, acc[4 * qq + 1] * scale_out, acc[4 * qq + 2] * scale_out,
                               acc[4 * qq + 3] * scale_out);
        float* yp = Y + (size_t)row * ldy + c0;
        if (EPI == SEPI_BIAS) {
          const float4 b = ld4(aux0 + c0);
          v = add4(v, b);
        } else if (EPI == SEPI_DACT) {
          const float4 hp = ld4(aux0 + (size_t)row * NF + c0);
          v = make_float4(v.x * dsilu_f(hp.x), v.y * dsilu_f(hp.y), v.z * dsilu_f(hp.z), v.w * dsilu_f(hp.w));
        } else if (EPI == SEPI_ACC) {
          v = add4(v, ld4(yp));
        } else if (EPI == SEPI_GF) {
          const float4 ga = ld4(aux1 + (size_t)(row / 3) * NF + c0), qv = ld4(aux2 + (size_t)row * NF + c0);
          v = fma4(ga, qv, v);
          if (aux0) v = add4(v, ld4(aux0 + (size_t)row * NF + c0));
        }
        st4(yp, v);
      }
    }
  }
}

__global__ void __launch_bounds__(SM_THREADS, 1) small_step_kernel(const SmallArgs A) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int N = A.N, E = A.E, L = A.L, P = E >> 1;
  const int c2 = 2 * lane;   // this lane's two features in the row phases

  // ---- embedding: atom_node = Embedding[z]; m of layer 0 from its per-element table
  for (int t = threadIdx.x; t < N * (NF / 4); t += SM_THREADS) {
    const int i = t / (NF / 4), c = t % (NF / 4);
    const size_t zi = (size_t)clamp_species(A.z[i]);
    reinterpret_cast<float4*>(A.a0)[t] = reinterpret_cast<const float4*>(A.emb + zi * NF)[c];
    reinterpret_cast<float4*>(A.m[0])[t] = reinterpret_cast<const float4*>(A.m_tab + zi * NF)[c];
  }
  __syncthreads();

  const float* a_in = A.a0;
  const float* f_in = nullptr;
  for (int l = 0; l < L; ++l) {
    const bool has_f = l > 0;
    // ---- messages + invariant aggregation (edge.hip:msg_fwd_kernel): wave per receiver row
    for (int i = wave; i < N; i += SM_WAVES) {
      const float2 mi = ld2(A.m[l] + (size_t)i * NF + c2);
      float2 acc = make_float2(0.f, 0.f);
      const int beg = A.row_ptr[i], end = A.row_ptr[i + 1];
      for (int e = beg; e < end; ++e) {
        const int j = A.col[e];
        const int2 gx = A.xg[e];
        const FilterW fw = filter_weights(__int_as_float(gx.y));
        const float* tp = A.ftab[l] + (size_t)gx.x * NF + c2;
        float2 eps = ld2(tp) * fw.w[0];
        eps = fma2(ld2(tp + NF), fw.w[1], eps);
        eps = fma2(ld2(tp + 2 * NF), fw.w[2], eps);
        eps = fma2(ld2(tp + 3 * NF), fw.w[3], eps);
        const float2 v = eps * mi * ld2(A.m[l] + (size_t)j * NF + c2);
        if (j > i) st2(A.msg[l] + (size_t)A.pid[e] * NF + c2, v);
        acc = acc + v;
      }
      st2(A.a_mid[l] + (size_t)i * NF + c2, ld2(a_in + (size_t)i * NF + c2) + acc);
    }
    __syncthreads();
    // ---- equiv_message1 / 2, first linears (pre-activations kept for the adjoint)
    wg_gemm<SPRO_NONE, SEPI_STORE>(A.msg[l], NF, A.img[l][IMG_EQ1_0], A.h1[l], NF, P, nullptr, nullptr, nullptr);
    if (has_f) wg_gemm<SPRO_NONE, SEPI_STORE>(A.msg[l], NF, A.img[l][IMG_EQ2_0], A.h2[l], NF, P, nullptr, nullptr, nullptr);
    __syncthreads();
    wg_gemm<SPRO_ACT, SEPI_STORE>(A.h1[l], NF, A.img[l][IMG_EQ1_2], A.phi1[l], NF, P, nullptr, nullptr, nullptr);
    if (has_f) wg_gemm<SPRO_ACT, SEPI_STORE>(A.h2[l], NF, A.img[l][IMG_EQ2_2], A.phi2[l], NF, P, nullptr, nullptr, nullptr);
    __syncthreads();
    // ---- equivariant messages + aggregation (edge.hip:force_fwd_kernel)
    for (int i = wave; i < N; i += SM_WAVES) {
      float2 acc[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) acc[k] = has_f ? ld2(f_in + ((size_t)i * 3 + k) * NF + c2) : make_float2(0.f, 0.f);
      const int beg = A.row_ptr[i], end = A.row_ptr[i + 1];
      for (int e = beg; e < end; ++e) {
        if (A.xg[e].x == FT_ZERO_ROW) continue;   // a candidate outside the cutoff (reused list): contributes nothing
        const float4 g = reinterpret_cast<const float4*>(A.geo)[e];
        const size_t p = (size_t)A.pid[e];
        const float2 v1 = ld2(A.phi1[l] + p * NF + c2);
        acc[0] = fma2(v1, g.x, acc[0]);
        acc[1] = fma2(v1, g.y, acc[1]);
        acc[2] = fma2(v1, g.z, acc[2]);
        if (has_f) {
          const int j = A.col[e];
          const float2 v2 = ld2(A.phi2[l] + p * NF + c2);
#pragma unroll
          for (int k = 0; k < 3; ++k) acc[k] = fma2(v2, ld2(f_in + ((size_t)j * 3 + k) * NF + c2), acc[k]);
        }
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) st2(A.f_out[l] + ((size_t)i * 3 + k) * NF + c2, acc[k]);
    }
    __syncthreads();
    // ---- equiv_update: q = f W_u^T over the 3N rows; energy update a_out = a_mid + sum_k f_k * q_k
    wg_gemm<SPRO_NONE, SEPI_STORE>(A.f_out[l], NF, A.img[l][IMG_UPDATE], A.q[l], NF, 3 * N, nullptr, nullptr, nullptr);
    __syncthreads();
    for (int i = wave; i < N; i += SM_WAVES) {
      float2 s = ld2(A.a_mid[l] + (size_t)i * NF + c2);
#pragma unroll
      for (int k = 0; k < 3; ++k)
        s = fma2(ld2(A.f_out[l] + ((size_t)i * 3 + k) * NF + c2), ld2(A.q[l] + ((size_t)i * 3 + k) * NF + c2), s);
      st2(A.a_out[l] + (size_t)i * NF + c2, s);
    }
    __syncthreads();
    // ---- the next layer's message_nodepart, or the first two linears of the energy head
    const bool last = l + 1 == L;
    float* hn = last ? A.e1 : A.hn[l + 1];
    float* mo = last ? A.e2 : A.m[l + 1];
    wg_gemm<SPRO_NONE, SEPI_BIAS>(A.a_out[l], NF, last ? A.img_head[IMG_HEAD0] : A.img[l + 1][IMG_NODE0], hn, NF, N,
                                  last ? A.head0_b : A.node0_b[l + 1], nullptr, nullptr);
    __syncthreads();
    wg_gemm<SPRO_ACT, SEPI_BIAS>(hn, NF, last ? A.img_head[IMG_HEAD2] : A.img[l + 1][IMG_NODE2], mo, NF, N,
                                 last ? A.head2_b : A.node2_b[l + 1], nullptr, nullptr);
    __syncthreads();
    a_in = A.a_out[l];
    f_in = A.f_out[l];
  }

  // ---- energy head tail + seed of the reverse sweep (edge.hip:head_out_kernel, mol_energy_kernel)
  for (int i = wave; i < N; i += SM_WAVES) {
    const float2 hv = ld2(A.e2 + (size_t)i * NF + c2);
    const float2 w = ld2(A.w4 + c2);
    const float s = wave_sum(fmaf(silu_f(hv.x), w.x, silu_f(hv.y) * w.y));
    const long zi = clamp_species(A.z[i]);
    const float sc = A.scale ? A.scale[zi] : 1.0f;
    const float sh = A.shift ? A.shift[zi] : 0.0f;
    if (lane == 0) A.atom_energy[i] = fmaf(s + A.b4[0], sc, sh);
    if (A.forces) st2(A.g_e + (size_t)i * NF + c2, make_float2(sc * w.x * dsilu_f(hv.x), sc * w.y * dsilu_f(hv.y)));
  }
  __syncthreads();
  for (int b = wave; b < A.B; b += SM_WAVES) {
    double s = 0.0;
    for (int i = A.mol_ptr[b] + lane; i < A.mol_ptr[b + 1]; i += 64) s += (double)A.atom_energy[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, WAVE);
    if (lane == 0) A.energy[b] = (float)s;
  }
  if (A.atom_node_out || A.force_node_out) {
    for (int t = threadIdx.x; t < N * (NF / 4); t += SM_THREADS)
      if (A.atom_node_out) reinterpret_cast<float4*>(A.atom_node_out)[t] = reinterpret_cast<const float4*>(A.a_out[L - 1])[t];
    for (int t = threadIdx.x; t < 3 * N * (NF / 4); t += SM_THREADS)
      if (A.force_node_out) reinterpret_cast<float4*>(A.force_node_out)[t] = reinterpret_cast<const float4*>(A.f_out[L - 1])[t];
  }
  if (!A.forces) return;

  // ================================================================== reverse sweep
  // head adjoint: g_e1 = (g_e2 H2) * silu'(e1) (written over e1);  g_a = g_e1 H0
  wg_gemm<SPRO_NONE, SEPI_DACT>(A.g_e, NF, A.img_head[IMG_HEAD2_T], A.e1, NF, N, A.e1, nullptr, nullptr);
  __syncthreads();
  wg_gemm<SPRO_NONE, SEPI_STORE>(A.e1, NF, A.img_head[IMG_HEAD0_T], A.g_a, NF, N, nullptr, nullptr, nullptr);
  __syncthreads();
  int pp = 0;
  const float* G_f = nullptr;   // dE/d f_out of the current layer from above (NULL = 0 at the top)
  for (int l = L - 1; l >= 0; --l) {
    const bool has_f = l > 0;
    float* scratch = A.g_f[pp];          // free until force_bwd of this layer writes it (never the buffer G_f points at)
    // ---- update adjoint: gf = G_f + g_a * q + (g_a * f) W_u      (newtonnet.py:230 a += sum_k f_k * (f_k W_u^T))
    for (int t = threadIdx.x; t < 3 * N * (NF / 4); t += SM_THREADS) {
      const int row = t / (NF / 4), c = t % (NF / 4);
      const float4 ga = reinterpret_cast<const float4*>(A.g_a)[(size_t)(row / 3) * (NF / 4) + c];
      reinterpret_cast<float4*>(scratch)[t] = mul4(ga, reinterpret_cast<const float4*>(A.f_out[l])[t]);
    }
    __syncthreads();
    wg_gemm<SPRO_NONE, SEPI_GF>(scratch, NF, A.img[l][IMG_UPDATE_T], A.gf, NF, 3 * N, G_f, A.g_a, A.q[l]);
    __syncthreads();
    // ---- adjoint of the equivariant messages (edge.hip:force_bwd_kernel)
    const float* f_prev = has_f ? A.f_out[l - 1] : nullptr;
    float* g_fin = A.g_f[pp];
    float* g_u = A.g_u + (size_t)4 * l * E;
    for (int i = wave; i < N; i += SM_WAVES) {
      float2 gfi[3], fi[3], acc[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        gfi[k] = ld2(A.gf + ((size_t)i * 3 + k) * NF + c2);
        acc[k] = gfi[k];
        fi[k] = has_f ? ld2(f_prev + ((size_t)i * 3 + k) * NF + c2) : make_float2(0.f, 0.f);
      }
      const int beg = A.row_ptr[i], end = A.row_ptr[i + 1];
      for (int e = beg; e < end; ++e) {
        const int j = A.col[e];
        const size_t p = (size_t)A.pid[e];
        const bool inside = A.xg[e].x != FT_ZERO_ROW;
        if (!inside) {
          if (lane == 0) reinterpret_cast<float4*>(g_u)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (j > i) {
            st2(A.g_phi1 + p * NF + c2, make_float2(0.f, 0.f));
            if (has_f) st2(A.g_phi2 + p * NF + c2, make_float2(0.f, 0.f));
          }
          continue;
        }
        const float4 g = reinterpret_cast<const float4*>(A.geo)[e];
        const float2 v1 = ld2(A.phi1[l] + p * NF + c2);
        float2 gfj[3];
        if (has_f || j > i) {
#pragma unroll
          for (int k = 0; k < 3; ++k) gfj[k] = ld2(A.gf + ((size_t)j * 3 + k) * NF + c2);
        }
        if (has_f) {
          const float2 v2 = ld2(A.phi2[l] + p * NF + c2);
#pragma unroll
          for (int k = 0; k < 3; ++k) acc[k] = fma2(v2, gfj[k], acc[k]);
        }
        if (j > i) {   // this row owns the pair: adjoints of the shared phi rows, both directions (u_ji = -u_ij)
          float2 gp1 = make_float2((gfi[0].x - gfj[0].x) * g.x, (gfi[0].y - gfj[0].y) * g.x);
          gp1 = make_float2(fmaf(gfi[1].x - gfj[1].x, g.y, gp1.x), fmaf(gfi[1].y - gfj[1].y, g.y, gp1.y));
          gp1 = make_float2(fmaf(gfi[2].x - gfj[2].x, g.z, gp1.x), fmaf(gfi[2].y - gfj[2].y, g.z, gp1.y));
          st2(A.g_phi1 + p * NF + c2, gp1);
          if (has_f) {
            float2 gp2 = make_float2(0.f, 0.f);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              gp2 = fma2(gfi[k], ld2(f_prev + ((size_t)j * 3 + k) * NF + c2), gp2);
              gp2 = fma2(gfj[k], fi[k], gp2);
            }
            st2(A.g_phi2 + p * NF + c2, gp2);
          }
        }
        const float s0 = wave_sum(fmaf(gfi[0].x, v1.x, gfi[0].y * v1.y));
        const float s1 = wave_sum(fmaf(gfi[1].x, v1.x, gfi[1].y * v1.y));
        const float s2 = wave_sum(fmaf(gfi[2].x, v1.x, gfi[2].y * v1.y));
        if (lane == 0) reinterpret_cast<float4*>(g_u)[e] = make_float4(s0, s1, s2, 0.f);
      }
      if (has_f) {
#pragma unroll
        for (int k = 0; k < 3; ++k) st2(g_fin + ((size_t)i * 3 + k) * NF + c2, acc[k]);
      }
    }
    __syncthreads();
    // ---- edge-MLP adjoints: t_k = (g_phi_k V_k2) * silu'(h_k) (written over h_k);  g_msg = t_1 V_10 + t_2 V_20
    wg_gemm<SPRO_NONE, SEPI_DACT>(A.g_phi1, NF, A.img[l][IMG_EQ1_2_T], A.h1[l], NF, P, A.h1[l], nullptr, nullptr);
    if (has_f) wg_gemm<SPRO_NONE, SEPI_DACT>(A.g_phi2, NF, A.img[l][IMG_EQ2_2_T], A.h2[l], NF, P, A.h2[l], nullptr, nullptr);
    __syncthreads();
    wg_gemm<SPRO_NONE, SEPI_STORE>(A.h1[l], NF, A.img[l][IMG_EQ1_0_T], A.g_msg, NF, P, nullptr, nullptr, nullptr);
    if (has_f) {
      __syncthreads();
      wg_gemm<SPRO_NONE, SEPI_ACC>(A.h2[l], NF, A.img[l][IMG_EQ2_0_T], A.g_msg, NF, P, nullptr, nullptr, nullptr);
    }
    __syncthreads();
    // ---- message adjoint (edge.hip:msg_bwd_kernel): g_m (layers > 0), g_x carried by the pair's owner
    float* g_x = A.g_x + (size_t)l * E;
    for (int i = wave; i < N; i += SM_WAVES) {
      const float2 mi = ld2(A.m[l] + (size_t)i * NF + c2);
      const float2 gai = ld2(A.g_a + (size_t)i * NF + c2);
      float2 acc = make_float2(0.f, 0.f);
      const int beg = A.row_ptr[i], end = A.row_ptr[i + 1];
      for (int e = beg; e < end; ++e) {
        const int j = A.col[e];
        if (j < i && !has_f) {
          if (lane == 0) g_x[e] = 0.f;
          continue;
        }
        const size_t p = (size_t)A.pid[e];
        const int2 gx = A.xg[e];
        const float2 mj = ld2(A.m[l] + (size_t)j * NF + c2);
        const float2 G = ld2(A.g_msg + p * NF + c2) + gai + ld2(A.g_a + (size_t)j * NF + c2);
        const FilterW fw = filter_weights(__int_as_float(gx.y));
        float2 eps;
        if (j > i) {
          const float* node = A.ftab[l] + (size_t)(gx.x + 1) * NF + c2;
          const float2 t0 = ld2(node), s0 = ld2(node + FT_PLANE), d0 = ld2(node + 2 * FT_PLANE), d1 = ld2(node + 2 * FT_PLANE + NF);
          eps = fma2(d1, fw.c, fma2(d0, fw.b, fma2(s0, fw.a, t0)));
          const float2 deps = fma2(d1, fw.dc, fma2(d0, fw.db, s0 * fw.da));
          const float2 gm = G * mi * mj * deps;
          const float gxs = wave_sum(gm.x + gm.y);
          if (lane == 0) g_x[e] = gxs;
        } else {
          const float* tp = A.ftab[l] + (size_t)gx.x * NF + c2;
          eps = ld2(tp) * fw.w[0];
          eps = fma2(ld2(tp + NF), fw.w[1], eps);
          eps = fma2(ld2(tp + 2 * NF), fw.w[2], eps);
          eps = fma2(ld2(tp + 3 * NF), fw.w[3], eps);
          if (lane == 0) g_x[e] = 0.f;
        }
        acc = fma2(G * eps, mj, acc);
      }
      if (has_f) st2(A.g_m + (size_t)i * NF + c2, acc);
    }
    __syncthreads();
    // ---- message_nodepart adjoint of this layer: g_hn = (g_m W2) * silu'(hn) (written over hn);  g_a += g_hn W0
    if (has_f) {
      wg_gemm<SPRO_NONE, SEPI_DACT>(A.g_m, NF, A.img[l][IMG_NODE2_T], A.hn[l], NF, N, A.hn[l], nullptr, nullptr);
      __syncthreads();
      wg_gemm<SPRO_NONE, SEPI_ACC>(A.hn[l], NF, A.img[l][IMG_NODE0_T], A.g_a, NF, N, nullptr, nullptr, nullptr);
      __syncthreads();
    }
    G_f = g_fin;
    pp ^= 1;
  }
  // ---- geometry adjoint -> forces (edge.hip:edge_gd_kernel, force_out_kernel)
  for (int e = threadIdx.x; e < E; e += SM_THREADS) {
    float gx = 0.f, gu0 = 0.f, gu1 = 0.f, gu2 = 0.f;
    for (int l = 0; l < L; ++l) {
      gx += A.g_x[(size_t)l * E + e];
      const float4 v = reinterpret_cast<const float4*>(A.g_u)[(size_t)l * E + e];
      gu0 += v.x;
      gu1 += v.y;
      gu2 += v.z;
    }
    const float4 g = reinterpret_cast<const float4*>(A.geo)[e];
    const float ir = 1.0f / g.w;
    const float dot = gu0 * g.x + gu1 * g.y + gu2 * g.z;
    const float a = gx * A.inv_rc - dot * ir;
    reinterpret_cast<float4*>(A.g_d)[e] = make_float4(fmaf(a, g.x, gu0 * ir), fmaf(a, g.y, gu1 * ir), fmaf(a, g.z, gu2 * ir), 0.f);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < N; i += SM_THREADS) {
    float fx = 0.f, fy = 0.f, fz = 0.f;
    for (int e = A.row_ptr[i]; e < A.row_ptr[i + 1]; ++e) {
      const float4 a = reinterpret_cast<const float4*>(A.g_d)[e];
      const float4 b = reinterpret_cast<const float4*>(A.g_d)[A.rev[e]];
      fx -= (a.x - b.x);
      fy -= (a.y - b.y);
      fz -= (a.z - b.z);
    }
    A.forces[3 * (size_t)i] = fx;
    A.forces[3 * (size_t)i + 1] = fy;
    A.forces[3 * (size_t)i + 2] = fz;
  }
}

int launch_small_step(const SmallArgs& a, hipStream_t s) {
  ScopedTimer t0(TC_OTHER, s);
  small_step_kernel<<<1, SM_THREADS, 0, s>>>(a);
  LAUNCH_CHECK();
  return 0;
}
