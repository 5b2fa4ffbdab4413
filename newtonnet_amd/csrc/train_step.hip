// The two halves of a training step as single C calls: the per-stage entry points of include/newtonnet_hip.h strung together
// exactly as tests/tangent_ref.py states the algorithm (sweeps 1-2 = values, 3-4 = tangents + weight gradients; csrc/train.hip has
// the kernels and the derivation).  Host code only: every line is a call to an exported stage; the buffers belong to the caller.
#include <string.h>

#include "common.h"

int launch_layer_norm_fwd(float* a, const float* gamma, const float* beta, int n_atoms, float* xhat, float* rstd, hipStream_t s);
int launch_layer_norm_bwd(float* g_a, const float* gamma, const float* xhat, const float* rstd, int n_atoms, hipStream_t s);
int launch_layer_norm_tan_fwd(float* da, const float* xhat, const float* rstd, const float* gamma, int n_atoms, float* dxhat,
                              float* drstd, hipStream_t s);
int launch_layer_norm_tan_bwd(const float* gy, float* dga, const float* xhat, const float* rstd, const float* dxhat,
                              const float* drstd, const float* gamma, int n_atoms, float* row_w, float* row_b, hipStream_t s);

// edge.hip: the launchers the inference step uses (pair_ptr: per-row pair counts; mol_ptr: batches of small molecules)
int launch_force_fwd(bool has_f, const float* phi1, const float* phi2, const float* geo, const int* row_ptr, const int* col,
                     const int* pid, const float* f_in, float* f_out, int n_atoms, const int* xg, hipStream_t s, const int* pair_ptr,
                     const int* mol_ptr, int n_mol);
int launch_force_bwd(bool has_f, const float* gf, const float* phi1, const float* phi2, const float* geo, const int* row_ptr,
                     const int* col, const int* pid, const float* f_in, float* g_h12, float* g_u, float* g_fin, int n_atoms,
                     const int* xg, hipStream_t s, const int* pair_ptr, const int* rev);
int launch_msg_bwd(const float* g_msg, const float* g_a, const float* m, const int* xg, const float* table, const int* row_ptr,
                   const int* col, const int* pid, float* g_m, float* g_x, int n_atoms, bool need_gm, hipStream_t s,
                   const int* pair_ptr, const int* mol_ptr, int n_mol);

#define TS_TRY(x)          \
  do {                     \
    int _r = (x);          \
    if (_r) return _r;     \
  } while (0)

static nnhip_mlp_desc mlp_desc(int mode, const float* X, int ldx, const float* W1, const float* W2, float* H, float* Y, int M,
                               int act, const void* img1 = nullptr, const void* img2 = nullptr) {
  nnhip_mlp_desc d;
  memset(&d, 0, sizeof(d));
  d.X = X;
  d.ldx = ldx;
  d.W1 = W1;
  d.W2 = W2;
  d.H = H;
  d.ldh = NF;
  d.Y = Y;
  d.ldy = NF;
  d.M = M;
  d.mode = mode;
  d.activation = act;
  d.W1_image = img1;     // (both NULL when the split-f16 form is off: train_images)
  d.W2_image = img2;
  return d;
}
// an edge MLP (equiv_message1/2, their adjoints and tangents) in the step's compute mode
static nnhip_mlp_desc edge_desc(bool bf16, int mode, const float* X, int ldx, const float* W1, const float* W2, float* H, float* Y, int M,
                                int act, const void* img1, const void* img2) {
  nnhip_mlp_desc d = mlp_desc(mode, X, ldx, W1, W2, H, Y, M, act, img1, img2);
  d.precision = bf16 ? 1 : 0;
  return d;
}
// bf16 compute mode of the step (torch.autocast(bfloat16): nnhip_train_ws.bf16_wgrad == 1): the edge MLPs of all four sweeps take
// bf16 operands next to the weight-gradient products; node-level kernels and everything elementwise stay fp32-grade
static bool train_bf16(const nnhip_model* model, const nnhip_train_ws* w) {
  return w->bf16_wgrad == 1 && split_products_enabled() && model->activation == NNHIP_ACT_SILU && w->himg[0] != nullptr;
}
static int run1(const nnhip_mlp_desc& d, void* s) { return d.M > 0 ? nnhip_mlp128_ex(&d, s) : NNHIP_OK; }
static int run2(const nnhip_mlp_desc& a, const nnhip_mlp_desc& b, void* s) { return a.M > 0 ? nnhip_mlp128_pair_ex(&a, &b, s) : NNHIP_OK; }

// the split-f16 product form of the row-local kernels: SiLU models, images present in the workspace, NNHIP_MLP_SPLIT != 0
static bool train_images(const nnhip_model* model, const nnhip_train_ws* w) {
  return split_products_enabled() && model->activation == NNHIP_ACT_SILU && w->himg[0] != nullptr;
}
#define LIMG(l, k) (img_on ? w->wimg[l][k] : nullptr)
#define HIMG(k) (img_on ? w->himg[k] : nullptr)

static int node_fwd_any(bool img_on, const nnhip_train_ws* w, int l, bool to_head, const float* f, const float* a_mid, const float* Wu,
                        float* q, float* a_out, const float* W0, const float* b0, const float* W2, const float* b2, float* hn,
                        float* m, int N, int act, void* s) {
  if (!img_on) return nnhip_node_fwd(f, a_mid, Wu, q, a_out, W0, b0, W2, b2, hn, m, N, act, s);
  NodeFwdArgs na;
  memset(&na, 0, sizeof(na));
  na.f = f;
  na.a_mid = a_mid;
  na.Wu = Wu;
  na.q = q;
  na.a_out = a_out;
  na.W0 = W0;
  na.b0 = b0;
  na.W2 = W2;
  na.b2 = b2;
  na.hn = hn;
  na.m = m;
  na.N = N;
  na.act = act;
  NodeImages im;
  memset(&im, 0, sizeof(im));
  im.Wu = (const char*)w->wimg[l][IMG_UPDATE];
  im.W0 = (const char*)(to_head ? w->himg[IMG_HEAD0] : w->wimg[l + 1][IMG_NODE0]);
  im.W2 = (const char*)(to_head ? w->himg[IMG_HEAD2] : w->wimg[l + 1][IMG_NODE2]);
  return launch_node_fwd_split(na, im, (hipStream_t)s);
}
// update adjoint only (the node-MLP adjoint of the training sweeps is a separate MODE_TAN launch that keeps its hidden product)
static int node_bwd_update_any(bool img_on, const nnhip_train_ws* w, int l, float* g_a, const float* f, const float* q, const float* G_f,
                               const float* WuT, float* gf, int N, int act, void* s) {
  if (!img_on) return nnhip_node_bwd(nullptr, nullptr, nullptr, nullptr, g_a, 0, f, q, G_f, WuT, gf, N, act, s);
  NodeBwdArgs nb;
  memset(&nb, 0, sizeof(nb));
  nb.g_a = g_a;
  nb.f = f;
  nb.q = q;
  nb.G_f = G_f;
  nb.WuT = WuT;
  nb.gf = gf;
  nb.N = N;
  nb.act = act;
  NodeImages im;
  memset(&im, 0, sizeof(im));
  im.WuT = (const char*)w->wimg[l][IMG_UPDATE_T];
  return launch_node_bwd_split(nb, im, (hipStream_t)s);
}

// value reverse sweep, split form: node-MLP / head adjoint of the upper level (keeps its hidden product T, reads the running g_a
// from `ga_in`, writes `ga_out`) + update adjoint of layer `lo` in ONE launch (node128s.hip:node_bwd_split_kernel)
static int node_bwd_fused(const nnhip_train_ws* w, const float* g_top, const float* h_top, const void* img_w2t, const void* img_w0t,
                          float* T, const float* ga_in, float* ga_out, int lo, const float* G_f, int N, int act, void* s) {
  NodeBwdArgs nb;
  memset(&nb, 0, sizeof(nb));
  nb.g_top = g_top;
  nb.h_top = h_top;
  nb.W2T = nb.W0T = w->wT[lo][6];   // (non-NULL: the split kernel takes its weights from the images below)
  nb.g_a = ga_out;
  nb.g_a_in = ga_in;
  nb.acc_ga = ga_in != nullptr;
  nb.T = T;
  nb.f = w->f_out[lo];
  nb.q = w->q[lo];
  nb.G_f = G_f;
  nb.WuT = w->wT[lo][6];
  nb.gf = w->gf[lo];
  nb.N = N;
  nb.act = act;
  NodeImages im;
  memset(&im, 0, sizeof(im));
  im.W2T = (const char*)img_w2t;
  im.W0T = (const char*)img_w0t;
  im.WuT = (const char*)w->wimg[lo][IMG_UPDATE_T];
  return launch_node_bwd_split(nb, im, (hipStream_t)s);
}

static int check(const nnhip_model* model, const nnhip_train_ws* w, const char* who) {
  if (!model || !w || w->n_layers != model->n_layers || w->n_layers < 1 || w->n_layers > NNHIP_MAX_LAYERS || w->n_atoms < 0 ||
      w->n_edges < 0 || (w->n_edges & 1) || model->n_features != NF || !w->rbf || !w->drbf) {
    nnhip_set_error("%s: bad arguments", who);
    return NNHIP_E_INVALID;
  }
  const bool ln = model->layer[0].ln_w != nullptr;
  for (int l = 0; l < model->n_layers; ++l) {
    const nnhip_layer_params& lp = model->layer[l];
    if ((lp.ln_w != nullptr) != ln || (lp.ln_w != nullptr) != (lp.ln_b != nullptr)) {
      nnhip_set_error("%s: layer_norm must be on for every interaction layer or for none", who);
      return NNHIP_E_UNSUPPORTED;
    }
    if (ln && !(w->ln_xhat[l] && w->ln_rstd[l] && w->ln_dxhat[l] && w->ln_drstd[l] && w->ln_gy[l] && w->ln_row_w[l] && w->ln_row_b[l])) {
      nnhip_set_error("%s: layer_norm=True needs the ln_* buffers of nnhip_train_ws", who);
      return NNHIP_E_INVALID;
    }
  }
  return NNHIP_OK;
}
// layer_norm=True: the node-level stages run unfused (update | LayerNorm | next message_nodepart) -- the fused row-local
// launches of node128s.hip have no LayerNorm stage -- with the LayerNorm value / tangent kernels between them
static bool has_ln(const nnhip_model* model) { return model->layer[0].ln_w != nullptr; }

extern "C" size_t nnhip_train_ws_bytes(void) { return sizeof(nnhip_train_ws); }

extern "C" int nnhip_train_values(const nnhip_model* model, const nnhip_train_ws* w, void* s) {
  TS_TRY(check(model, w, "nnhip_train_values"));
  const int N = w->n_atoms, E = w->n_edges, B = w->n_mol, L = w->n_layers, P = E / 2, act = model->activation;
  if (N == 0) return NNHIP_OK;
  const bool img_on = train_images(model, w);
  const bool ln = has_ln(model);
  const bool node_img = img_on && !ln;
  const bool mol_forms = (w->flags & 1) && w->pair_ptr && w->mol_ptr;   // every molecule fits the molecule-resident kernels
  const bool bf = train_bf16(model, w);
  // parameter-only data of this step: transposed weights, radial-filter tables
  {
    const float* src[40];
    float* dst[40];
    int c = 0;
    for (int l = 0; l < L; ++l) {
      const nnhip_layer_params& lp = model->layer[l];
      const float* ws_[7] = {lp.node0_w, lp.node2_w, lp.eq1_0_w, lp.eq1_2_w, lp.eq2_0_w, lp.eq2_2_w, lp.update_w};
      for (int k = 0; k < 7; ++k) {
        if (c == 40) {
          TS_TRY(nnhip_transpose128(src, dst, c, s));
          c = 0;
        }
        src[c] = ws_[k];
        dst[c++] = w->wT[l][k];
      }
    }
    if (c + 2 > 40) {
      TS_TRY(nnhip_transpose128(src, dst, c, s));
      c = 0;
    }
    src[c] = model->head0_w;
    dst[c++] = w->headT[0];
    src[c] = model->head2_w;
    dst[c++] = w->headT[1];
    TS_TRY(nnhip_transpose128(src, dst, c, s));
    if (train_images(model, w)) {   // split-f16 images of every weight and transpose (the weights change every step)
      const float* isrc[IMG_PER_LAYER * NNHIP_MAX_LAYERS + IMG_HEAD_COUNT];
      void* idst[IMG_PER_LAYER * NNHIP_MAX_LAYERS + IMG_HEAD_COUNT];
      int n = 0;
      for (int l = 0; l < L; ++l) {
        const nnhip_layer_params& lp = model->layer[l];
        const float* m_[IMG_PER_LAYER];
        m_[IMG_UPDATE] = lp.update_w;
        m_[IMG_NODE0] = lp.node0_w;
        m_[IMG_NODE2] = lp.node2_w;
        m_[IMG_UPDATE_T] = w->wT[l][6];
        m_[IMG_NODE0_T] = w->wT[l][0];
        m_[IMG_NODE2_T] = w->wT[l][1];
        m_[IMG_EQ1_0] = lp.eq1_0_w;
        m_[IMG_EQ1_2] = lp.eq1_2_w;
        m_[IMG_EQ2_0] = lp.eq2_0_w;
        m_[IMG_EQ2_2] = lp.eq2_2_w;
        m_[IMG_EQ1_0_T] = w->wT[l][2];
        m_[IMG_EQ1_2_T] = w->wT[l][3];
        m_[IMG_EQ2_0_T] = w->wT[l][4];
        m_[IMG_EQ2_2_T] = w->wT[l][5];
        for (int k = 0; k < IMG_PER_LAYER; ++k) {
          isrc[n] = m_[k];
          idst[n++] = w->wimg[l][k];
        }
      }
      const float* h_[IMG_HEAD_COUNT] = {model->head0_w, model->head2_w, w->headT[0], w->headT[1]};
      for (int k = 0; k < IMG_HEAD_COUNT; ++k) {
        isrc[n] = h_[k];
        idst[n++] = w->himg[k];
      }
      TS_TRY(nnhip_weight_images(isrc, idst, n, s));
      if (train_bf16(model, w)) {     // the eight edge-MLP images of every layer once more, in the bf16 format
        int nb_ = 0;
        for (int l = 0; l < L; ++l)
          for (int k = IMG_EQ1_0; k <= IMG_EQ2_2_T; ++k) {
            isrc[nb_] = isrc[l * IMG_PER_LAYER + k];
            idst[nb_++] = w->wimg[l][k];
          }
        TS_TRY(nnhip_weight_images_bf16(isrc, idst, nb_, s));
      }
    }
    const float* ew[NNHIP_MAX_LAYERS];
    for (int l = 0; l < L; ++l) ew[l] = model->layer[l].edge_w;
    TS_TRY(nnhip_filter_tables(ew, w->ftab, L, model->frequencies, model->n_basis, model->envelope, s));
  }
  // ---- sweep 1: forward
  TS_TRY(nnhip_embed(w->z, model->node_embedding, N, w->a0, s));
  {
    const nnhip_layer_params& l0 = model->layer[0];
    nnhip_mlp_desc d = mlp_desc(MODE_FWD, w->a0, NF, l0.node0_w, l0.node2_w, w->hn[0], w->m[0], N, act, LIMG(0, IMG_NODE0),
                                LIMG(0, IMG_NODE2));
    d.b1 = l0.node0_b;
    d.b2 = l0.node2_b;
    TS_TRY(run1(d, s));
  }
  const float* a_in = w->a0;
  const float* f_in = nullptr;
  for (int l = 0; l < L; ++l) {
    const nnhip_layer_params& lp = model->layer[l];
    TS_TRY(nnhip_message_fwd(w->m[l], w->xg, w->ftab[l], w->row_ptr, w->col, w->pid, a_in, w->msg[l], w->a_mid[l], N, s));
    const nnhip_mlp_desc d1 = edge_desc(bf, MODE_FWD, w->msg[l], NF, lp.eq1_0_w, lp.eq1_2_w, w->h1[l], w->phi1[l], P, act,
                                       LIMG(l, IMG_EQ1_0), LIMG(l, IMG_EQ1_2));
    if (l > 0)
      TS_TRY(run2(d1, edge_desc(bf, MODE_FWD, w->msg[l], NF, lp.eq2_0_w, lp.eq2_2_w, w->h2[l], w->phi2[l], P, act, LIMG(l, IMG_EQ2_0),
                               LIMG(l, IMG_EQ2_2)), s));
    else
      TS_TRY(run1(d1, s));
    // (round 6: as the inference step -- per-row pair counts, and for batches of small molecules the molecule-resident form)
    TS_TRY(launch_force_fwd(f_in != nullptr, w->phi1[l], w->phi2[l], w->geo, w->row_ptr, w->col, w->pid, f_in, w->f_out[l], N, w->xg,
                            (hipStream_t)s, w->pair_ptr, mol_forms ? w->mol_ptr : nullptr, B));
    if (ln) {   // update | LayerNorm (in place; x_hat, 1/sigma kept) | next message_nodepart or head, unfused
      const bool lastl = l + 1 == L;
      TS_TRY(nnhip_node_fwd(w->f_out[l], w->a_mid[l], lp.update_w, w->q[l], w->a_out[l], nullptr, nullptr, nullptr, nullptr, nullptr,
                            nullptr, N, act, s));
      TS_TRY(launch_layer_norm_fwd(w->a_out[l], lp.ln_w, lp.ln_b, N, w->ln_xhat[l], w->ln_rstd[l], (hipStream_t)s));
      nnhip_mlp_desc d = mlp_desc(MODE_FWD, w->a_out[l], NF, lastl ? model->head0_w : model->layer[l + 1].node0_w,
                                  lastl ? model->head2_w : model->layer[l + 1].node2_w, lastl ? w->e1 : w->hn[l + 1],
                                  lastl ? w->e2 : w->m[l + 1], N, act);
      d.b1 = lastl ? model->head0_b : model->layer[l + 1].node0_b;
      d.b2 = lastl ? model->head2_b : model->layer[l + 1].node2_b;
      TS_TRY(run1(d, s));
    } else if (l + 1 < L) {
      const nnhip_layer_params& nx = model->layer[l + 1];
      TS_TRY(node_fwd_any(img_on, w, l, false, w->f_out[l], w->a_mid[l], lp.update_w, w->q[l], w->a_out[l], nx.node0_w, nx.node0_b,
                          nx.node2_w, nx.node2_b, w->hn[l + 1], w->m[l + 1], N, act, s));
    } else {
      TS_TRY(node_fwd_any(img_on, w, l, true, w->f_out[l], w->a_mid[l], lp.update_w, w->q[l], w->a_out[l], model->head0_w,
                          model->head0_b, model->head2_w, model->head2_b, w->e1, w->e2, N, act, s));
    }
    a_in = w->a_out[l];
    f_in = w->f_out[l];
  }
  TS_TRY(nnhip_head_out(w->e2, model->head4_w, model->head4_b, model->scale, model->shift, w->z, w->mol_ptr, N, B, act,
                        w->atom_energy, w->g_e2, w->energy, s));
  // ---- sweep 2: reverse (seed 1)
  // (layer_norm: the gradient at the LayerNorm OUTPUT is kept in ln_gy -- the tangent sweep differentiates the LayerNorm adjoint
  // at it -- and GA[l] holds the gradient at the pre-norm row, which is what the update / message adjoints of layer l consume)
  auto ln_adjoint = [&](int l) -> int {
    HIP_TRY(hipMemcpyAsync(w->ln_gy[l], w->GA[l], sizeof(float) * (size_t)N * NF, hipMemcpyDeviceToDevice, (hipStream_t)s));
    return launch_layer_norm_bwd(w->GA[l], model->layer[l].ln_w, w->ln_xhat[l], w->ln_rstd[l], N, (hipStream_t)s);
  };
  if (node_img) {
    TS_TRY(node_bwd_fused(w, w->g_e2, w->e1, w->himg[IMG_HEAD2_T], w->himg[IMG_HEAD0_T], w->t_e1, nullptr, w->GA[L - 1], L - 1,
                          nullptr, N, act, s));
  } else {
    nnhip_mlp_desc d = mlp_desc(MODE_TAN, w->g_e2, NF, w->headT[1], w->headT[0], w->e1, w->GA[L - 1], N, act);
    d.T = w->t_e1;
    TS_TRY(run1(d, s));
    if (ln) TS_TRY(ln_adjoint(L - 1));
    TS_TRY(node_bwd_update_any(false, w, L - 1, w->GA[L - 1], w->f_out[L - 1], w->q[L - 1], nullptr, w->wT[L - 1][6], w->gf[L - 1], N,
                               act, s));
  }
  int pp = 0;
  for (int l = L - 1; l >= 0; --l) {
    const float* f_prev = l > 0 ? w->f_out[l - 1] : nullptr;
    float* Gf = w->Gf[pp];
    TS_TRY(launch_force_bwd(f_prev != nullptr, w->gf[l], w->phi1[l], w->phi2[l], w->geo, w->row_ptr, w->col, w->pid, f_prev,
                            w->g_h12[l], w->g_u + (size_t)4 * l * E, Gf, N, w->xg, (hipStream_t)s, w->pair_ptr, w->rev));
    nnhip_mlp_desc d1 = edge_desc(bf, MODE_TAN, w->g_h12[l], 2 * NF, w->wT[l][3], w->wT[l][2], w->h1[l], w->g_msg[l], P, act,
                                 LIMG(l, IMG_EQ1_2_T), LIMG(l, IMG_EQ1_0_T));
    d1.T = w->t1[l];
    if (l > 0) {
      nnhip_mlp_desc d2 = edge_desc(bf, MODE_TAN, w->g_h12[l] + NF, 2 * NF, w->wT[l][5], w->wT[l][4], w->h2[l], w->g_msg[l], P, act,
                                   LIMG(l, IMG_EQ2_2_T), LIMG(l, IMG_EQ2_0_T));
      d2.T = w->t2[l];
      d2.accumulate = 1;
      TS_TRY(run2(d1, d2, s));
    } else {
      TS_TRY(run1(d1, s));
    }
    TS_TRY(launch_msg_bwd(w->g_msg[l], w->GA[l], w->m[l], w->xg, w->ftab[l], w->row_ptr, w->col, w->pid,
                          l > 0 ? w->g_m[l] : nullptr, w->g_x + (size_t)l * E, N, l > 0, (hipStream_t)s, w->pair_ptr,
                          mol_forms ? w->mol_ptr : nullptr, B));
    if (l > 0 && node_img) {
      TS_TRY(node_bwd_fused(w, w->g_m[l], w->hn[l], w->wimg[l][IMG_NODE2_T], w->wimg[l][IMG_NODE0_T], w->t_n[l], w->GA[l],
                            w->GA[l - 1], l - 1, Gf, N, act, s));
    } else if (l > 0) {
      HIP_TRY(hipMemcpyAsync(w->GA[l - 1], w->GA[l], sizeof(float) * (size_t)N * NF, hipMemcpyDeviceToDevice, (hipStream_t)s));
      nnhip_mlp_desc d = mlp_desc(MODE_TAN, w->g_m[l], NF, w->wT[l][1], w->wT[l][0], w->hn[l], w->GA[l - 1], N, act);
      d.T = w->t_n[l];
      d.accumulate = 1;
      TS_TRY(run1(d, s));
      if (ln) TS_TRY(ln_adjoint(l - 1));
      TS_TRY(node_bwd_update_any(false, w, l - 1, w->GA[l - 1], w->f_out[l - 1], w->q[l - 1], Gf, w->wT[l - 1][6], w->gf[l - 1], N, act,
                                 s));
    }
    pp ^= 1;
  }
  return nnhip_edge_embed_bwd(w->g_x, w->g_u, w->geo, w->disp, w->pos, w->cell, w->row_ptr, w->col, w->rev, w->mol_ptr, N, E, B,
                              L, model->cutoff, w->g_d, w->forces, nullptr, s);
}

extern "C" int nnhip_train_grads(const nnhip_model* model, const nnhip_train_ws* w, const float* g_energy,
                                 const float* g_forces, void* s) {
  return nnhip_train_grads_seeded(model, w, g_energy, g_forces, nullptr, nullptr, s);
}

// seed_a [N][F] / seed_f [N][3][F] (either may be NULL): dL/d atom_node, dL/d force_node of loss terms that read the final node
// states directly (the direct_force head, csrc/heads.hip).  They enter the epsilon-part of the reverse sweep at its top: that
// sweep is the tangent of the value adjoint and is linear in its seeds, so every weight-gradient product below picks up the
// plain back-propagation of these terms next to the tangent-over-reverse terms of the energy / gradient-force loss.
extern "C" int nnhip_train_grads_seeded(const nnhip_model* model, const nnhip_train_ws* w, const float* g_energy,
                                        const float* g_forces, const float* seed_a, const float* seed_f, void* s) {
  TS_TRY(check(model, w, "nnhip_train_grads"));
  if (!g_energy || !g_forces) {
    nnhip_set_error("nnhip_train_grads: bad arguments");
    return NNHIP_E_INVALID;
  }
  const int N = w->n_atoms, E = w->n_edges, L = w->n_layers, P = E / 2, act = model->activation;
  if (N == 0) return NNHIP_OK;
  const bool img_on = train_images(model, w);
  const bool ln = has_ln(model);
  const bool node_img = img_on && !ln;
  const bool bf = train_bf16(model, w);
  // ---- sweep 3: tangent forward along v = -dL/dF
  TS_TRY(nnhip_edge_tangent_geom(g_forces, -1.0f, w->edge_index, w->geo, E, model->cutoff, w->tgeo, s));
  for (int l = 0; l < L; ++l) {
    const nnhip_layer_params& lp = model->layer[l];
    const bool first = l == 0;
    TS_TRY(nnhip_message_tan_fwd(w->m[l], first ? nullptr : w->dm[l], w->xg, w->tgeo, w->ftab[l], w->row_ptr, w->col, w->pid,
                                 first ? nullptr : w->da_out[l - 1], w->dmsg[l], w->da_mid, N, s));
    nnhip_mlp_desc d1 = edge_desc(bf, MODE_TAN, w->dmsg[l], NF, lp.eq1_0_w, lp.eq1_2_w, w->h1[l], w->dphi1[l], P, act, LIMG(l, IMG_EQ1_0),
                                 LIMG(l, IMG_EQ1_2));
    d1.T = w->dh1[l];
    if (!first) {
      nnhip_mlp_desc d2 = edge_desc(bf, MODE_TAN, w->dmsg[l], NF, lp.eq2_0_w, lp.eq2_2_w, w->h2[l], w->dphi2[l], P, act,
                                   LIMG(l, IMG_EQ2_0), LIMG(l, IMG_EQ2_2));
      d2.T = w->dh2[l];
      TS_TRY(run2(d1, d2, s));
    } else {
      TS_TRY(run1(d1, s));
    }
    TS_TRY(nnhip_force_message_tan_fwd(w->phi1[l], w->dphi1[l], w->phi2[l], w->dphi2[l], w->geo, w->tgeo, w->xg, w->row_ptr,
                                       w->col, w->pid, first ? nullptr : w->f_out[l - 1], first ? nullptr : w->df_out[l - 1],
                                       w->df_out[l], N, s));
    if (node_img) {   // equiv_update tangent + energy-update tangent + tangent of the next message_nodepart / head: one launch
      const bool last = l + 1 == L;
      NodeTanFwdArgs a;
      memset(&a, 0, sizeof(a));
      a.df = w->df_out[l];
      a.f = w->f_out[l];
      a.q = w->q[l];
      a.da_mid = w->da_mid;
      a.hn = last ? w->e1 : w->hn[l + 1];
      a.dq = w->dq[l];
      a.da_out = w->da_out[l];
      a.T = last ? w->de1 : w->dhn[l + 1];
      a.Y = last ? w->de2 : w->dm[l + 1];
      a.N = N;
      NodeImages im;
      memset(&im, 0, sizeof(im));
      im.Wu = (const char*)w->wimg[l][IMG_UPDATE];
      im.W0 = (const char*)(last ? w->himg[IMG_HEAD0] : w->wimg[l + 1][IMG_NODE0]);
      im.W2 = (const char*)(last ? w->himg[IMG_HEAD2] : w->wimg[l + 1][IMG_NODE2]);
      TS_TRY(launch_node_tan_fwd_split(a, im, (hipStream_t)s));
      continue;
    }
    TS_TRY(nnhip_linear128(w->df_out[l], NF, lp.update_w, w->dq[l], NF, nullptr, nullptr, 0, 3 * N, PRO_NONE, EPI_STORE, s));
    TS_TRY(nnhip_update_tan_fwd(w->da_mid, w->f_out[l], w->df_out[l], w->q[l], w->dq[l], N, w->da_out[l], s));
    if (ln)
      TS_TRY(launch_layer_norm_tan_fwd(w->da_out[l], w->ln_xhat[l], w->ln_rstd[l], lp.ln_w, N, w->ln_dxhat[l], w->ln_drstd[l],
                                       (hipStream_t)s));
    if (l + 1 < L) {
      const nnhip_layer_params& nx = model->layer[l + 1];
      nnhip_mlp_desc d = mlp_desc(MODE_TAN, w->da_out[l], NF, nx.node0_w, nx.node2_w, w->hn[l + 1], w->dm[l + 1], N, act);
      d.T = w->dhn[l + 1];
      TS_TRY(run1(d, s));
    } else {
      nnhip_mlp_desc d = mlp_desc(MODE_TAN, w->da_out[l], NF, model->head0_w, model->head2_w, w->e1, w->de2, N, act);
      d.T = w->de1;
      TS_TRY(run1(d, s));
    }
  }
  // ---- sweep 4: tangent reverse, seed tangent c = dL/dE
  TS_TRY(nnhip_head_seed_tan(w->e2, w->de2, model->head4_w, model->head4_b, model->scale, w->z, w->batch, g_energy, N, act,
                             w->dg_e2, w->w4row, w->scal, s));
  // split form: the tangent of a node-MLP (or head) adjoint and the tangent of the update adjoint BELOW it share one launch
  // (node128s.hip:node_tan_bwd_split_kernel): "A(level) + B(layer)"
  auto tan_bwd_fused = [&](const float* g_top, const float* h_top, const float* t2, const float* hd, float* G, int acc,
                           const void* img_w2t, const void* img_w0t, int lo, const float* dgf_in) -> int {
    NodeTanBwdArgs a;
    memset(&a, 0, sizeof(a));
    a.g_top = g_top;
    a.h_top = h_top;
    a.t2_top = t2;
    a.hd_top = hd;
    a.G = G;
    a.dga = w->dGA;
    a.acc_dga = acc;
    a.N = N;
    NodeImages im;
    memset(&im, 0, sizeof(im));
    im.W2T = (const char*)img_w2t;
    im.W0T = (const char*)img_w0t;
    if (lo >= 0) {
      a.ga = w->GA[lo];
      a.f = w->f_out[lo];
      a.df = w->df_out[lo];
      a.q = w->q[lo];
      a.dq = w->dq[lo];
      a.dgf_in = dgf_in;
      a.gq = w->gq[lo];
      a.dgq = w->dgq[lo];
      a.dgf = w->dgf;
      im.WuT = (const char*)w->wimg[lo][IMG_UPDATE_T];
    }
    return launch_node_tan_bwd_split(a, im, (hipStream_t)s);
  };
  if (seed_a)
    HIP_TRY(hipMemcpyAsync(w->dGA, seed_a, sizeof(float) * (size_t)N * NF, hipMemcpyDeviceToDevice, (hipStream_t)s));
  auto ln_tan_adjoint = [&](int l) -> int {   // dGA: tangent of the gradient at the LayerNorm output -> at the pre-norm row
    return launch_layer_norm_tan_bwd(w->ln_gy[l], w->dGA, w->ln_xhat[l], w->ln_rstd[l], w->ln_dxhat[l], w->ln_drstd[l],
                                     model->layer[l].ln_w, N, w->ln_row_w[l], w->ln_row_b[l], (hipStream_t)s);
  };
  if (node_img) {
    TS_TRY(tan_bwd_fused(w->dg_e2, w->e1, w->t_e1, w->de1, w->dg_e1, seed_a ? 1 : 0, w->himg[IMG_HEAD2_T], w->himg[IMG_HEAD0_T],
                         L - 1, seed_f));
  } else {
    nnhip_mlp_desc d = mlp_desc(MODE_TAN2, w->dg_e2, NF, w->headT[1], w->headT[0], w->e1, w->dGA, N, act);
    d.T2 = w->t_e1;
    d.Hd = w->de1;
    d.G = w->dg_e1;
    d.accumulate = seed_a ? 1 : 0;
    TS_TRY(run1(d, s));
    if (ln) TS_TRY(ln_tan_adjoint(L - 1));
  }
  const float* dGf = node_img ? nullptr : seed_f;
  int pp = 0;
  for (int l = L - 1; l >= 0; --l) {
    const bool first = l == 0;
    if (!node_img) {
      TS_TRY(nnhip_update_tan_bwd(w->GA[l], w->dGA, w->f_out[l], w->df_out[l], w->q[l], w->dq[l], dGf, N, w->gq[l], w->dgq[l],
                                  w->dgf, s));
      TS_TRY(nnhip_linear128(w->dgq[l], NF, w->wT[l][6], w->dgf, NF, nullptr, nullptr, 0, 3 * N, PRO_NONE, EPI_ACC, s));
    }
    float* nxt = w->dGf[pp];
    TS_TRY(nnhip_force_message_tan_bwd(w->gf[l], w->dgf, w->phi2[l], w->dphi2[l], w->geo, w->tgeo, w->xg, w->row_ptr, w->col,
                                       w->pid, first ? nullptr : w->f_out[l - 1], first ? nullptr : w->df_out[l - 1], w->dg_h12[l],
                                       first ? nullptr : nxt, N, s));
    nnhip_mlp_desc d1 = edge_desc(bf, MODE_TAN2, w->dg_h12[l], 2 * NF, w->wT[l][3], w->wT[l][2], w->h1[l], w->dg_msg, P, act,
                                 LIMG(l, IMG_EQ1_2_T), LIMG(l, IMG_EQ1_0_T));
    d1.T2 = w->t1[l];
    d1.Hd = w->dh1[l];
    d1.G = w->dg_h1[l];
    if (!first) {
      nnhip_mlp_desc d2 = edge_desc(bf, MODE_TAN2, w->dg_h12[l] + NF, 2 * NF, w->wT[l][5], w->wT[l][4], w->h2[l], w->dg_msg, P, act,
                                   LIMG(l, IMG_EQ2_2_T), LIMG(l, IMG_EQ2_0_T));
      d2.T2 = w->t2[l];
      d2.Hd = w->dh2[l];
      d2.G = w->dg_h2[l];
      d2.accumulate = 1;
      TS_TRY(run2(d1, d2, s));
    } else {
      TS_TRY(run1(d1, s));
    }
    TS_TRY(nnhip_message_tan_bwd(w->g_msg[l], w->dg_msg, w->GA[l], w->dGA, w->m[l], first ? nullptr : w->dm[l], w->xg, w->tgeo,
                                 w->ftab[l], w->row_ptr, w->col, w->pid, w->dg_m[l], w->g_eps[l], w->dg_eps[l], N, s));
    if (node_img) {
      TS_TRY(tan_bwd_fused(w->dg_m[l], w->hn[l], first ? nullptr : w->t_n[l], first ? nullptr : w->dhn[l], w->dg_hn[l], 1,
                           w->wimg[l][IMG_NODE2_T], w->wimg[l][IMG_NODE0_T], l - 1, first ? nullptr : nxt));
    } else {
      nnhip_mlp_desc d = mlp_desc(MODE_TAN2, w->dg_m[l], NF, w->wT[l][1], w->wT[l][0], w->hn[l], w->dGA, N, act);
      d.T2 = first ? w->zeros_nf : w->t_n[l];
      d.Hd = first ? w->zeros_nf : w->dhn[l];
      d.G = w->dg_hn[l];
      d.accumulate = 1;
      TS_TRY(run1(d, s));
      if (ln && !first) TS_TRY(ln_tan_adjoint(l - 1));
    }
    dGf = nxt;
    pp ^= 1;
  }
  // ---- weight gradients: one batched split-K launch + its reduction, column sums, per-element sums
  TS_TRY(nnhip_pair_rbf(w->rbf, w->drbf, w->tgeo, w->edge_index, w->pid, E, w->n_basis, w->rb, s));
  TS_TRY(nnhip_wgrad_batch(w->probs, w->n_probs, w->chunks, w->slabs, w->bf16_wgrad, P, s));
  TS_TRY(nnhip_colsum_batch(w->sums, w->n_sums, w->cs_scratch, s));
  TS_TRY(nnhip_species_sum(w->dGA, NF, NF, w->z, N, w->sp_scratch, w->g_embedding, 0, NF, NF, nullptr, 0, 0, 0, nullptr, 0, s));
  return nnhip_species_sum(w->scal, 4, 4, w->z, N, w->sp_scratch, w->g_scale, 0, 1, 1, w->g_shift, 1, 1, 1, w->g_head4_b, 2, s);
}
