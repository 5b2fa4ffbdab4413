// Shared device helpers for the gfx950 kernels of libnewtonnet_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/newtonnet_hip.h"

// Every compile-time switch that changes RESULTS (the ablations of tools/ablate_*.sh) may only be defined in a tooling build
// (build.sh adds -DNNHIP_TOOLING to any build with extra flags; nnhip_build_flags() bit 0 reports it and newtonnet_amd.hip
// refuses to load such a library unless NNHIP_ALLOW_TOOLING_LIB=1).
#if (defined(EDGE_ABL_SELF) || defined(EDGE_ABL_PAIR) || defined(EDGE_ABL_TABLE) || defined(EDGE_ABL_STORE) ||              \
     defined(EDGE_ABL_NO_TABLE) || defined(EDGE_ABL_HALF_TABLE) || defined(EDGE_ABL_NOTAB) || defined(EDGE_ABL_NOMJ) || defined(MLPS_ABL_NO_H) || defined(NS_ABL_NO_Q) || defined(NS_ABL_HOT_W) || defined(MLPS_ABL_X) || defined(ABL_NO_SILU) || defined(ABL_NO_STORE) ||                  \
     (defined(LIN_ABLATE_NO_LOAD) && LIN_ABLATE_NO_LOAD) || (defined(LIN_ABLATE_NO_STORE) && LIN_ABLATE_NO_STORE) ||         \
     (defined(LIN_ABLATE_NO_MFMA) && LIN_ABLATE_NO_MFMA) || (defined(LIN_ABLATE_NO_LDS) && LIN_ABLATE_NO_LDS)) &&            \
    !defined(NNHIP_TOOLING)
#error "ablation switches produce wrong results: build through csrc/build.sh (it marks the library with -DNNHIP_TOOLING)"
#endif

#define NF NNHIP_F    // 128 features: one wave = 64 lanes x float2
#define NB NNHIP_NB   // 20 radial basis functions (default; 1..NNHIP_MAX_NB supported)
#define WAVE 64
#ifndef FT_G
#define FT_G 3072            // radial-filter table: intervals on x = r/cutoff in [0, 1)
#endif
// Three planes T | S | D of FT_ROWS rows each; row = node + 1, nodes x_g = g / FT_G for g = -1 .. FT_G + 6 (zero beyond the
// cutoff).  A candidate edge at or beyond the cutoff (Verlet-skin lists, nnhip_edge_disp) carries the node index FT_ZERO_ROW:
// every row either stencil touches from there is all-zero (edge_common.h)
#define FT_ROWS (FT_G + 8)
#define FT_ZERO_ROW (FT_G + 3)
#define FT_PLANE ((size_t)FT_ROWS * NNHIP_F)

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- error plumbing (host) -------------------------------------------------
void nnhip_set_error(const char* fmt, ...);
#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      nnhip_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return NNHIP_E_HIP;                                                                  \
    }                                                                                      \
  } while (0)
#define LAUNCH_CHECK() HIP_TRY(hipGetLastError())

// ---- timers (host) ---------------------------------------------------------
enum { TC_EDGE = 0, TC_LIN = 1, TC_OTHER = 2, TC_EDGE_FWD_MSG = 3, TC_EDGE_FWD_FORCE = 4, TC_EDGE_BWD_FORCE = 5,
       TC_EDGE_BWD_MSG = 6, TC_GRAPH = 7, TC_MLP = 8, TC_LIN1 = 9, TC_WGRAD = 10, TC_MLP_ONEPASS = 11, TC_MOL_FWD = 12, TC_MOL_BWD = 13 };
struct ScopedTimer {
  int cls;
  hipStream_t s;
  hipEvent_t e0;
  bool on;
  ScopedTimer(int cls, hipStream_t s);
  ~ScopedTimer();
};

// ---- device helpers ----------------------------------------------------------
// v_exp_f32 + v_rcp_f32 (about 1 ulp each): well inside the fp32 tolerance of the path
#ifdef ABL_NO_SILU   // tooling only (wrong results): price the activation's VALU work
__device__ __forceinline__ float sigmoid_f(float x) { return 0.5f + 0.25f * x; }
#else
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
#endif
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_f(x); }
__device__ __forceinline__ float dsilu_f(float x) {
  const float s = sigmoid_f(x);
  return s * (1.0f + x * (1.0f - s));
}

// The other activations of the reference's factory (newtonnet/layers/activations.py:5-30; torch defaults: ELU alpha 1,
// LeakyReLU slope 0.01, Softplus beta 1 / threshold 20, exact-erf GELU, ShiftedSoftplus = softplus - ln 2).  SiLU is the
// default of every published config and keeps its own fast path (NNHIP_ACT_SILU = 0: the kernels branch on it once,
// wave-uniformly); the rest use the accurate libm forms -- nobody tunes for them.
__device__ __forceinline__ float act_f(float x, int a) {
  switch (a) {
    case NNHIP_ACT_RELU: return fmaxf(x, 0.f);
    case NNHIP_ACT_ELU: return x > 0.f ? x : expm1f(x);
    case NNHIP_ACT_LEAKY_RELU: return x > 0.f ? x : 0.01f * x;
    case NNHIP_ACT_TANH: return tanhf(x);
    case NNHIP_ACT_SIGMOID: return 1.0f / (1.0f + expf(-x));
    case NNHIP_ACT_SOFTPLUS: return x > 20.f ? x : log1pf(expf(x));
    case NNHIP_ACT_GELU: return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
    case NNHIP_ACT_SSP: return (x > 20.f ? x : log1pf(expf(x))) - 0.69314718055994531f;
    default: return silu_f(x);
  }
}
__device__ __forceinline__ float dact_f(float x, int a) {
  switch (a) {
    case NNHIP_ACT_RELU: return x > 0.f ? 1.f : 0.f;
    case NNHIP_ACT_ELU: return x > 0.f ? 1.f : expf(x);
    case NNHIP_ACT_LEAKY_RELU: return x > 0.f ? 1.f : 0.01f;
    case NNHIP_ACT_TANH: {
      const float t = tanhf(x);
      return 1.f - t * t;
    }
    case NNHIP_ACT_SIGMOID: {
      const float g = 1.0f / (1.0f + expf(-x));
      return g * (1.f - g);
    }
    case NNHIP_ACT_SOFTPLUS:
    case NNHIP_ACT_SSP: return x > 20.f ? 1.f : 1.0f / (1.0f + expf(-x));
    case NNHIP_ACT_GELU:
      return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
    default: return dsilu_f(x);
  }
}
// second derivatives: the tangent (forward-mode) sweeps of training differentiate act'(h) once more (train.hip)
__device__ __forceinline__ float d2silu_f(float x) {
  const float s = sigmoid_f(x);
  return s * (1.0f - s) * (2.0f + x * (1.0f - 2.0f * s));
}
__device__ __forceinline__ float d2act_f(float x, int a) {
  switch (a) {
    case NNHIP_ACT_RELU:
    case NNHIP_ACT_LEAKY_RELU: return 0.f;
    case NNHIP_ACT_ELU: return x > 0.f ? 0.f : expf(x);
    case NNHIP_ACT_TANH: {
      const float t = tanhf(x);
      return -2.f * t * (1.f - t * t);
    }
    case NNHIP_ACT_SIGMOID: {
      const float g = 1.0f / (1.0f + expf(-x));
      return g * (1.f - g) * (1.f - 2.f * g);
    }
    case NNHIP_ACT_SOFTPLUS:
    case NNHIP_ACT_SSP: {
      if (x > 20.f) return 0.f;
      const float g = 1.0f / (1.0f + expf(-x));
      return g * (1.f - g);
    }
    case NNHIP_ACT_GELU: return (2.f - x * x) * 0.3989422804014327f * expf(-0.5f * x * x);
    default: return d2silu_f(x);
  }
}
__device__ __forceinline__ float act_any(float x, int a) { return a == NNHIP_ACT_SILU ? silu_f(x) : act_f(x, a); }
__device__ __forceinline__ float dact_any(float x, int a) { return a == NNHIP_ACT_SILU ? dsilu_f(x) : dact_f(x, a); }
__device__ __forceinline__ float d2act_any(float x, int a) { return a == NNHIP_ACT_SILU ? d2silu_f(x) : d2act_f(x, a); }
// apply to a register block: one wave-uniform test keeps the SiLU loop exactly as it was
#define NN_ACT_BLOCK(v, n, a)                                            \
  do {                                                                   \
    if ((a) == NNHIP_ACT_SILU) {                                         \
      _Pragma("unroll") for (int k_ = 0; k_ < (n); ++k_)(v)[k_] = silu_f((v)[k_]);  \
    } else {                                                             \
      _Pragma("unroll") for (int k_ = 0; k_ < (n); ++k_)(v)[k_] = act_f((v)[k_], (a)); \
    }                                                                    \
  } while (0)
#define NN_DACT_MUL_BLOCK(v, pre, n, a)                                  \
  do {                                                                   \
    if ((a) == NNHIP_ACT_SILU) {                                         \
      _Pragma("unroll") for (int k_ = 0; k_ < (n); ++k_)(v)[k_] *= dsilu_f((pre)[k_]);  \
    } else {                                                             \
      _Pragma("unroll") for (int k_ = 0; k_ < (n); ++k_)(v)[k_] *= dact_f((pre)[k_], (a)); \
    }                                                                    \
  } while (0)

// Wave-wide sum on the DPP path (register-to-register cross-lane moves, a few cycles each; __shfl_xor compiles to
// ds_bpermute_b32, an LDS-pipe round trip of ~100 cycles per step, and six of those in a row were a third of the per-edge
// latency chain of the message / force adjoints).  quad_perm swaps -> row_half_mirror -> row_mirror leave the 16-lane row
// sums in every lane; row_bcast15 / row_bcast31 (gfx9 DPP) fold the four rows into lane 63.
#define NN_DPP_ADD(v, ctrl, row_mask)                                                                            \
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, row_mask, 0xF, false))
__device__ __forceinline__ float wave_sum_lane63(float v) {   // the total is valid in lane 63 only
  NN_DPP_ADD(v, 0xB1, 0xF);    // quad_perm [1,0,3,2]
  NN_DPP_ADD(v, 0x4E, 0xF);    // quad_perm [2,3,0,1]
  NN_DPP_ADD(v, 0x141, 0xF);   // row_half_mirror
  NN_DPP_ADD(v, 0x140, 0xF);   // row_mirror
  NN_DPP_ADD(v, 0x142, 0xA);   // row_bcast15 into rows 1 and 3
  NN_DPP_ADD(v, 0x143, 0xC);   // row_bcast31 into rows 2 and 3
  return v;
}
__device__ __forceinline__ float half_sum_top(float v) {      // sums of lanes 0-31 / 32-63, valid in lanes 31 / 63 only
  NN_DPP_ADD(v, 0xB1, 0xF);
  NN_DPP_ADD(v, 0x4E, 0xF);
  NN_DPP_ADD(v, 0x141, 0xF);
  NN_DPP_ADD(v, 0x140, 0xF);
  NN_DPP_ADD(v, 0x142, 0xA);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {          // the total in every lane (as a scalar broadcast)
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_sum_lane63(v)), 63));
}

// Block b is observed to run on XCD b % 8 (each XCD has a private 4 MiB L2).  Give every XCD a
// contiguous range of tiles so that the rows a molecule's atoms gather from stay in one L2.
// Bijective for any n_blocks.  A speed choice only: correctness never depends on placement.
__device__ __forceinline__ int xcd_tile(int b, int n_blocks) {
  const int q = n_blocks >> 3, r = n_blocks & 7;
  const int x = b & 7, k = b >> 3;
  const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + k;
}

// 16-byte row pieces.  The texture addresser spends ~16 cycles on a wave64 vector-memory instruction whatever its width
// (PMC: TA_BUSY 90-100 % in the edge kernels while they moved 8 bytes per lane), so the edge kernels use float4 per lane.
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 ld4_nt(const float* p) {
  typedef float v4 __attribute__((ext_vector_type(4)));
  const v4 v = __builtin_nontemporal_load(reinterpret_cast<const v4*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st4_nt(float* p, float4 v) {
  typedef float v4 __attribute__((ext_vector_type(4)));
  v4 w;
  w.x = v.x;
  w.y = v.y;
  w.z = v.z;
  w.w = v.w;
  __builtin_nontemporal_store(w, reinterpret_cast<v4*>(p));
}
__device__ __forceinline__ float4 fma4(float4 a, float4 b, float4 c) {
  return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}
__device__ __forceinline__ float4 fma4(float4 a, float s, float4 c) {
  return make_float4(fmaf(a.x, s, c.x), fmaf(a.y, s, c.y), fmaf(a.z, s, c.z), fmaf(a.w, s, c.w));
}
__device__ __forceinline__ float4 mul4(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 mul4(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 sub4(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float dot4(float4 a, float4 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w))); }
// lanes 0-31 receive the value held by lane + 32 (folding the two half-wave partial sums of a row)
__device__ __forceinline__ float4 upper_half(float4 v) {
  return make_float4(__shfl_down(v.x, 32, WAVE), __shfl_down(v.y, 32, WAVE), __shfl_down(v.z, 32, WAVE), __shfl_down(v.w, 32, WAVE));
}

__device__ __forceinline__ float2 ld2(const float* p) { return *reinterpret_cast<const float2*>(p); }
__device__ __forceinline__ void st2(float* p, float2 v) { *reinterpret_cast<float2*>(p) = v; }
// streaming variants (edge tensors are touched once per kernel: keep them from evicting the gathered node rows in L2)
__device__ __forceinline__ float2 ld2_nt(const float* p) {
  typedef float v2 __attribute__((ext_vector_type(2)));
  const v2 v = __builtin_nontemporal_load(reinterpret_cast<const v2*>(p));
  return make_float2(v.x, v.y);
}
__device__ __forceinline__ void st2_nt(float* p, float2 v) {
  typedef float v2 __attribute__((ext_vector_type(2)));
  v2 w;
  w.x = v.x;
  w.y = v.y;
  __builtin_nontemporal_store(w, reinterpret_cast<v2*>(p));
}
__device__ __forceinline__ float2 operator*(float2 a, float2 b) { return make_float2(a.x * b.x, a.y * b.y); }
__device__ __forceinline__ float2 operator*(float2 a, float s) { return make_float2(a.x * s, a.y * s); }
__device__ __forceinline__ float2 operator+(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 fma2(float2 a, float2 b, float2 c) { return make_float2(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)); }
__device__ __forceinline__ float2 fma2(float2 a, float s, float2 c) { return make_float2(fmaf(a.x, s, c.x), fmaf(a.y, s, c.y)); }

// species index into the [119]-row tables: values outside 0..118 are flagged by check_species_kernel and raised by the host (possibly
// AFTER the step was queued: nnhip_energy_forces_dev) -- the kernels must never index with them
__device__ __forceinline__ long clamp_species(long z) { return z < 0 ? 0 : (z >= NNHIP_N_ELEMENTS ? NNHIP_N_ELEMENTS - 1 : z); }

// ---- dense 128x128 linear launcher (lin128.hip) ---------------------------------
enum { PRO_NONE = 0, PRO_SILU = 1 };
enum { EPI_STORE = 0, EPI_BIAS = 1, EPI_DSILU = 2, EPI_ACC = 3 };
struct LinGroup {
  const float* A;
  const float* W;     // [128 out][128 in] row-major
  float* C;
  const float* bias;  // EPI_BIAS: [128]
  const float* H;     // EPI_DSILU: C = acc * silu'(H[m][n])
};
struct LinArgs {
  LinGroup g[2];      // blockIdx.y selects (two independent linears of the same shape in one launch)
  int M;
  int lda, ldc, ldh;  // row strides in floats
  int act;            // NNHIP_ACT_* applied where PRO_SILU / EPI_DSILU say "SiLU" (0 = SiLU)
};
int launch_lin(int pro, int epi, const LinArgs& a, int groups, hipStream_t s);

// ---- fused two-layer edge MLP launcher (mlp128.hip) -------------------------------
// MODE_TAN  = MODE_BWD that also stores T (the stage-1 product before the act' factor) -- the tangent of the forward MLP
//             (T = d h) and, in train mode, the adjoint of the forward MLP with its intermediate kept (T = g_phi V2)
// MODE_TAN2 = tangent of the adjoint: stage 1  dT = X W1^T;  G = dT * act'(H) + T2 * act''(H) * Hd  (stored);  Y (+)= G W2^T
enum { MODE_FWD = 0, MODE_BWD = 1, MODE_TAN = 2, MODE_TAN2 = 3 };
struct MlpArgs {
  const float* X;   // [M][ldx]  stage-1 input (msg, or g_phi)
  const float* W1;  // [128][128] row-major, stage 1:  H^T = W1 . X^T
  const float* W2;  // [128][128] row-major, stage 2:  Y^T = W2 . act^T
  float* H;         // [M][ldh]  FWD: output (pre-activation);  BWD: input (pre-activation of the forward)
  float* Y;         // [M][ldy]  stage-2 output
  int M, ldx, ldh, ldy;
  const float* b1;  // optional biases (node MLP / energy head; forward mode, row-local small-M kernel only)
  const float* b2;
  int act;          // NNHIP_ACT_* (0 = SiLU)
  int h_frag;       // H is private scratch between a forward call and its adjoint (same M): the persistent kernel then keeps
                    // it in MFMA-fragment order ([tile][block][q][lane] float4: every access a contiguous 1 KiB, so the
                    // streaming stores write whole lines) in a region of pad32(M) x 128 floats; ldh is ignored
  // training modes (row pitch ldh, row-local kernel only):
  float* T;         // MODE_TAN: [M][ldh] out, stage-1 product before the act' factor
  const float* T2;  // MODE_TAN2: [M][ldh] in, the T of the value sweep
  const float* Hd;  // MODE_TAN2: [M][ldh] in, tangent of H
  float* G;         // MODE_TAN2: [M][ldh] out, the hidden adjoint's tangent
  // optional split-f16 images of W1 / W2 (node128s.hip:weight_image_kernel): the row-local kernel then takes the split-f16 form
  const char* W1_img;
  const char* W2_img;
  // bf16 compute mode (training under torch.autocast(bfloat16), BASELINE configs[2]): operands rounded to bf16, ONE
  // v_mfma_f32_32x32x16_bf16 per 16 k-values, fp32 accumulation; SiLU; the images (row-local form) must be WIMG_FMT_BF16
  int bf16;
  // optional: the TRUE row count on the device (a step queued before the host knows its edge count: nnhip_energy_forces_dev).
  // M then is the capacity the arrays, the layout and the grid are sized for; the kernels process min(M, *M_dev) rows.
  const int* M_dev;
};
// rows a kernel works on (uniform: a scalar load)
__device__ __forceinline__ int mlp_rows(const MlpArgs& a) { return a.M_dev ? min(a.M, *a.M_dev) : a.M; }
struct MlpPair {      // up to two MLPs over the same M rows, run back to back by one persistent launch (mlp128.hip)
  MlpArgs a[2];
  int n;
  int accum[2];       // stage-2 output is added to Y instead of overwriting it
};
int launch_mlp(int mode, bool accum, const MlpArgs& a, hipStream_t s);
int launch_mlp_pair(int mode, const MlpArgs& a0, bool accum0, const MlpArgs& a1, bool accum1, hipStream_t s);
// mlp128r.hip: both MLPs of a layer in one pass over the rows, weights resident in registers (large batches, FWD / BWD, images given)
bool mlp_regw_serves(int mode, const MlpPair& P);
int launch_mlp_regw(int mode, const MlpPair& P, hipStream_t s);

// ---- row-local fused node kernels (node128.hip) -------------------------------------
struct NodeFwdArgs {
  const float* f;      // [N][3][F] force_node after the edge phase (f')
  const float* a_mid;  // [N][F]
  const float* Wu;     // [F][F]
  float* q;            // [N][3][F] out
  float* a_out;        // [N][F] out
  const float *W0, *b0, *W2, *b2;  // next layer's message_nodepart (W0 == NULL: no next layer)
  float *hn, *m;       // [N][F] out (next layer)
  int N;
  int act;             // NNHIP_ACT_* of the fused message_nodepart / head MLP
};
struct NodeBwdArgs {
  const float* g_top;  // [N][F]  g_m of the upper layer (or g_e2 at the head)
  const float* h_top;  // [N][F]  hn of the upper layer (or e1): pre-activation for silu'
  const float *W2T, *W0T;  // transposed weights of the upper node MLP / head
  float* g_a;          // [N][F]  running dE/d atom_node  (acc_ga: g_a += ..., else g_a = ...)
  const float* f;      // [N][3][F] force_node of the lower layer after its edge phase
  const float* q;      // [N][3][F]
  const float* G_f;    // [N][3][F] dE/d f_out of the lower layer from above (NULL = 0)
  const float* WuT;    // [F][F] transposed equiv_update of the lower layer
  float* gf;           // [N][3][F] out
  int N;
  int acc_ga;
  int act;
  // training (split-f16 kernel only): keep the hidden product t = g_top W2 before the act' factor; read the running g_a from
  // another buffer than the one written (the value sweep keeps g_a of every layer)
  float* T;
  const float* g_a_in;
};
// tangent forms of the two node kernels (node128s.hip; training sweeps 3 and 4, csrc/train.hip)
struct NodeTanFwdArgs {
  const float *df, *f, *q;   // [N][3][F] tangent / value of force_node after the edge phase, equiv_update(force_node)
  const float* da_mid;       // [N][F]
  const float* hn;           // [N][F] pre-activation of the next message_nodepart / head (value sweep)
  float* dq;                 // [N][3][F] out
  float* da_out;             // [N][F] out
  float *T, *Y;              // [N][F] out: tangent of hn, tangent of the next m (or of e2)
  int N;
};
struct NodeTanBwdArgs {
  // part A (g_top != NULL): tangent of the upper node-MLP / head adjoint
  const float *g_top, *h_top, *t2_top, *hd_top;   // dg_m, hn, t_n, dhn (t2 / hd NULL = 0)
  float* G;                  // [N][F] out: tangent of the hidden adjoint
  float* dga;                // [N][F] running tangent of dE/d atom_node (acc_dga: +=, else =; part A absent: input only)
  int acc_dga;
  // part B (f != NULL): tangent of the lower layer's update adjoint
  const float *ga, *f, *df, *q, *dq, *dgf_in;
  float *gq, *dgq, *dgf;     // [N][3][F] out
  int N;
};
int launch_node_tan_fwd_split(const NodeTanFwdArgs& a, const struct NodeImages& im, hipStream_t s);
int launch_node_tan_bwd_split(const NodeTanBwdArgs& a, const struct NodeImages& im, hipStream_t s);
int launch_node_fwd(const NodeFwdArgs& a, hipStream_t s);
int launch_node_bwd(const NodeBwdArgs& a, hipStream_t s);
// split-f16 forms (node128s.hip): the weights come as prepared (hi, lo) f16 images instead of the fp32 pointers of the arguments
#define WIMG_BYTES (2 * NF * NF * 2 + 8)   // two 16-bit planes + the inverse scale + the format word (WIMG_FMT_*)
#define WIMG_FMT_SPLIT_F16 0               // hi / lo f16 planes of the matrix scaled by a power of two (fp32-grade products, 3 MFMAs)
#define WIMG_FMT_BF16 1                    // first plane = the matrix rounded to bf16, unscaled; second plane unused (1 MFMA)
#define WIMG_MAX_JOBS 40
struct WeightImageJobs {
  const float* src[WIMG_MAX_JOBS];
  char* dst[WIMG_MAX_JOBS];
  int fmt;                                  // WIMG_FMT_* of every image of the launch
};
// images kept per layer (prepared block / training workspace), and of the energy head
enum { IMG_UPDATE = 0, IMG_NODE0, IMG_NODE2, IMG_UPDATE_T, IMG_NODE0_T, IMG_NODE2_T, IMG_EQ1_0, IMG_EQ1_2, IMG_EQ2_0, IMG_EQ2_2,
       IMG_EQ1_0_T, IMG_EQ1_2_T, IMG_EQ2_0_T, IMG_EQ2_2_T, IMG_PER_LAYER };
enum { IMG_HEAD0 = 0, IMG_HEAD2, IMG_HEAD0_T, IMG_HEAD2_T, IMG_HEAD_COUNT };
struct NodeImages {
  const char *Wu, *W0, *W2;      // node_fwd: equiv_update, next message_nodepart / head (first, second linear)
  const char *W2T, *W0T, *WuT;   // node_bwd: their transposes
};
int launch_weight_images(const float* const* src, char* const* dst, int n, hipStream_t s, int fmt = WIMG_FMT_SPLIT_F16);
int launch_node_fwd_split(const NodeFwdArgs& a, const NodeImages& im, hipStream_t s);
int launch_node_bwd_split(const NodeBwdArgs& a, const NodeImages& im, hipStream_t s);
int launch_mlp_wide_split(int mode, bool accum, const MlpArgs& a, hipStream_t s);      // needs a.W1_img / a.W2_img, SiLU
int launch_mlp_wide_pair_split(int mode, const MlpPair& P, hipStream_t s);
int launch_lin_wide_split(const float* X, int ldx, const char* img, float* Y, int ldy, int M, bool acc, hipStream_t s);
bool split_products_enabled();   // mlp128.hip (NNHIP_MLP_SPLIT=0 turns every split-f16 kernel off)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
