// Training through the direct_force head (newtonnet/models/output.py:115-132, scalers.py:55-56; DirectForceLoss
// newtonnet/train/loss.py:41-47): the head's own first-order adjoint.  Unlike the gradient force, the direct force is an
// ordinary function of (atom_node, force_node): its loss back-propagates once.  This file produces
//   * the head's parameter-gradient operands (g_d3, g_pre2, g_pre1: rows of the batched weight-gradient launch and of the
//     column sums that nnhip_train_grads runs at the end of the step) and the scale gradient,
//   * the adjoint seeds dL/d atom_node, dL/d force_node, which nnhip_train_grads_seeded injects into the epsilon-part of the
//     reverse sweep (csrc/train_step.hip) -- that sweep is linear in its seeds, so the interaction layers' gradients of the
//     direct-force loss come out of the same launches as those of the energy / gradient-force loss.
// Forward: nnhip_direct_force (node128.hip) with the caller KEEPING its scratch = (pre1 | pre2 | d3), [3][N][F].
#include <string.h>

#include "common.h"

int launch_transposes(const float* const* src, float* const* dst, int count, hipStream_t s);

// out[i][k] = scale_i <d3[i], f[i][k]>  =>  g_d3[i] = scale_i sum_k g[i][k] f[i][k];  seed_f[i][k] = g[i][k] scale_i d3[i];
// dL/dscale[z_i] += sum_k g[i][k] <d3[i], f[i][k]>  (sc4[i][0], summed per element afterwards)
__global__ void __launch_bounds__(256)
direct_force_adj_kernel(const float* __restrict__ g_out /*[N][3]*/, const float* __restrict__ d3, const float* __restrict__ force_node,
                        const float* __restrict__ scale, const int64_t* __restrict__ z, int n_atoms, float* __restrict__ g_d3,
                        float* __restrict__ seed_f, float* __restrict__ sc4) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n_atoms) return;
  const int lane = threadIdx.x & 63;
  const float sc = scale ? scale[clamp_species(z[i])] : 1.0f;
  const float2 dv = ld2(d3 + (size_t)i * NF + 2 * lane);
  float2 acc = make_float2(0.f, 0.f);
  float raw = 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float g = g_out[(size_t)i * 3 + k];
    const float2 f = ld2(force_node + ((size_t)i * 3 + k) * NF + 2 * lane);
    acc = fma2(f, g * sc, acc);
    raw = fmaf(g, wave_sum(fmaf(dv.x, f.x, dv.y * f.y)), raw);
    st2(seed_f + ((size_t)i * 3 + k) * NF + 2 * lane, dv * (g * sc));
  }
  st2(g_d3 + (size_t)i * NF + 2 * lane, acc);
  if (lane < 4) sc4[(size_t)i * 4 + lane] = lane == 0 ? raw : 0.f;
}

__global__ void __launch_bounds__(256)
dact_mul_kernel(const float* __restrict__ t, const float* __restrict__ pre, int act, size_t n4, float* __restrict__ out) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n4) return;
  const float4 a = ld4(t + 4 * k), h = ld4(pre + 4 * k);
  st4(out + 4 * k, make_float4(a.x * dact_any(h.x, act), a.y * dact_any(h.y, act), a.z * dact_any(h.z, act), a.w * dact_any(h.w, act)));
}

extern "C" size_t nnhip_direct_force_bwd_work_floats(int32_t n_atoms) {
  return (size_t)4 * (n_atoms > 0 ? n_atoms : 0) * NF + (size_t)3 * NF * NF;
}

// keep = the scratch of the forward nnhip_direct_force call (pre1 | pre2 | d3).  work: nnhip_direct_force_bwd_work_floats(N)
// floats = g_d3 | g_pre2 | t1 | g_pre1 ([N][F] each, in this order: the caller's weight-gradient table points into them) and the
// three transposed weights.  seed_a [N][F], seed_f [N][3][F]: outputs.  g_scale: [119] gradient of scalers.k.scale.weight
// (NULL when the head has no scale); sc4 [N][4] + sp_scratch (nnhip_species_scratch_bytes(4)): its scratch.
extern "C" int nnhip_direct_force_bwd(const float* g_out, const float* force_node, const int64_t* z, const float* w0,
                                      const float* w2, const float* w4, const float* scale, int32_t activation, int32_t n_atoms,
                                      const float* keep, float* work, float* seed_a, float* seed_f, float* sc4, float* sp_scratch,
                                      float* g_scale, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (!g_out || !force_node || !z || !w0 || !w2 || !w4 || !keep || !work || !seed_a || !seed_f || !sc4 || n_atoms < 0 ||
      activation < NNHIP_ACT_SILU || activation > NNHIP_ACT_SSP || (g_scale && (!scale || !sp_scratch))) {
    nnhip_set_error("nnhip_direct_force_bwd: bad arguments");
    return NNHIP_E_INVALID;
  }
  if (n_atoms == 0) return NNHIP_OK;
  const size_t nf = (size_t)n_atoms * NF;
  const float *pre1 = keep, *pre2 = keep + nf, *d3 = keep + 2 * nf;
  float *g_d3 = work, *g_pre2 = work + nf, *t1 = work + 2 * nf, *g_pre1 = work + 3 * nf;
  float* wT = work + 4 * nf;   // w4^T | w2^T | w0^T
  {
    const float* src[3] = {w4, w2, w0};
    float* dst[3] = {wT, wT + NF * NF, wT + 2 * NF * NF};
    int rc = launch_transposes(src, dst, 3, s);
    if (rc) return rc;
  }
  direct_force_adj_kernel<<<cdiv(n_atoms, 4), 256, 0, s>>>(g_out, d3, force_node, scale, z, n_atoms, g_d3, seed_f, sc4);
  LAUNCH_CHECK();
  // g_pre2 = (g_d3 W4) * act'(pre2)            (d3 = act(pre2) W4^T + b4)
  LinArgs l;
  memset(&l, 0, sizeof(l));
  l.g[0] = {g_d3, wT, g_pre2, nullptr, pre2};
  l.M = n_atoms;
  l.lda = l.ldc = l.ldh = NF;
  l.act = activation;
  int rc = launch_lin(PRO_NONE, EPI_DSILU, l, 1, s);
  if (rc) return rc;
  // t1 = g_pre2 W2;  seed_a = (t1 * act'(pre1)) W0      (the adjoint form of the fused MLP kernel, hidden product kept)
  MlpArgs a;
  memset(&a, 0, sizeof(a));
  a.X = g_pre2;
  a.W1 = wT + NF * NF;
  a.W2 = wT + 2 * NF * NF;
  a.H = const_cast<float*>(pre1);
  a.Y = seed_a;
  a.T = t1;
  a.M = n_atoms;
  a.ldx = a.ldh = a.ldy = NF;
  a.act = activation;
  rc = launch_mlp(MODE_TAN, false, a, s);
  if (rc) return rc;
  dact_mul_kernel<<<cdiv((int)(nf / 4), 256), 256, 0, s>>>(t1, pre1, activation, nf / 4, g_pre1);
  LAUNCH_CHECK();
  if (g_scale)
    return nnhip_species_sum(sc4, 4, 4, z, n_atoms, sp_scratch, g_scale, 0, 1, 1, nullptr, 0, 0, 0, nullptr, 0, s);
  return NNHIP_OK;
}
