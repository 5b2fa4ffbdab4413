// Host-side orchestration of the hot path behind the C ABI (include/newtonnet_hip.h): workspace carving,
// the forward sweep and the analytic reverse sweep (energy -> forces), error strings, event timers.
//
// Mirrors NewtonNet.forward (newtonnet/models/newtonnet.py:74-104) for output_properties
// ['energy', 'gradient_force'] and replaces torch.autograd.grad (newtonnet/models/output.py:66-73) by explicit
// adjoint kernels.  Every launch goes to the caller's stream; nothing here synchronises.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "common.h"

// ---- pieces defined in the other translation units ---------------------------------------------------
int launch_filter_tables(const float* const* edge_w, float* const* tables, int n_layers, const float* freq, int nb,
                         int envelope, hipStream_t s);
int launch_msg_fwd(const float* m, const int* xg, const float* table, const int* row_ptr, const int* col,
                   const int* pid, const float* a_in, float* msg, float* a_mid, int n_atoms, hipStream_t s);
int launch_force_fwd(bool has_f, const float* phi1, const float* phi2, const float* geo, const int* row_ptr,
                     const int* col, const int* pid, const float* f_in, float* f_out, int n_atoms, const int* xg, hipStream_t s,
                     const int* pair_ptr = nullptr, const int* mol_ptr = nullptr, int n_mol = 0);
int launch_force_bwd(bool has_f, const float* gf, const float* phi1, const float* phi2, const float* geo,
                     const int* row_ptr, const int* col, const int* pid, const float* f_in, float* g_h12, float* g_u,
                     float* g_fin, int n_atoms, const int* xg, hipStream_t s, const int* pair_ptr = nullptr, const int* rev = nullptr);
int launch_msg_bwd(const float* g_msg, const float* g_a, const float* m, const int* xg, const float* table,
                   const int* row_ptr, const int* col, const int* pid, float* g_m, float* g_x, int n_atoms, bool need_gm,
                   hipStream_t s, const int* pair_ptr = nullptr, const int* mol_ptr = nullptr, int n_mol = 0);
int launch_geometry_bwd(const float* g_x, const float* g_u, const float* geo, const float* disp, const float* pos,
                        const float* cell, const int* row_ptr, const int* col, const int* rev, const int* mol_ptr,
                        int n_atoms, int n_edges, int n_mol, int n_layers, float cutoff, float* g_d, float* forces,
                        float* virial, hipStream_t s, bool small_molecules = false);
int launch_layer_norm_fwd(float* a, const float* gamma, const float* beta, int n_atoms, float* xhat, float* rstd,
                          hipStream_t s);
int launch_layer_norm_bwd(float* g_a, const float* gamma, const float* xhat, const float* rstd, int n_atoms,
                          hipStream_t s);
int launch_embed(const int64_t* z, const float* table, const float* m_table, int n_atoms, float* a0, float* m0,
                 hipStream_t s);
int launch_head_out(const float* e2, const float* w4, const float* b4, const float* scale, const float* shift,
                    const int64_t* z, const int* mol_ptr, int n_atoms, int n_mol, int act, float* atom_energy, float* g_e2,
                    float* energy, hipStream_t s, bool small_molecules = false);
int launch_transposes(const float* const* src, float* const* dst, int count, hipStream_t s);
// ---- errors ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void nnhip_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* nnhip_last_error(void) { return g_err; }
extern "C" int nnhip_version(void) { return 108; }   // 108: nnhip_spatial_order / nnhip_permute_rows / nnhip_edge_index_unpermute, the bf16 compute mode (nnhip_mlp_desc.precision, nnhip_weight_images_bf16), one-pass training forms of the edge MLPs, nnhip_train_ws.pair_ptr / .flags; 107: the molecule-resident fused edge phase (molfuse.hip, molfuse2.hip) and the single-launch small step (small.hip) removed -- measured slower than the row path, profiles/r06_fused_persistent_*.txt; 106: the molecule-resident fused edge phase (molfuse.hip, NNHIP_MOL_FUSED), timer classes 12 / 13; 105: molecule-resident force_fwd (status bit 8, flags); 104: the deferred step (nnhip_forward_dev, nnhip_graph_small_dev), split rows; 103: nnhip_prepare_check
// bit 0: tooling build (compiled with extra flags -- ablation / A-B switches); never loaded by the package by default
extern "C" int nnhip_build_flags(void) {
#ifdef NNHIP_TOOLING
  return 1;
#else
  return 0;
#endif
}

// Every form choice the library makes, as one JSON object (bench.py prints it in its line; tests pin the non-default forms through
// the switches it lists).  "env" holds the NNHIP_* switches that are SET in this process, whatever they say; the other keys are what
// the code does with them (thresholds are in atoms / molecules / 32-row tiles).
void edge_config(int* small_atoms, int* mol_min, int* wpr, int* mol_forms);   // edge.hip
int mlp_wide_max_tiles_silu();                                                 // mlp128.hip
extern "C" int nnhip_mlp_forms(void);
extern "C" int nnhip_graph_small_max_atoms(void);
extern "C" int nnhip_config(char* buf, size_t n) {
  if (!buf || n < 64) {
    nnhip_set_error("nnhip_config: buffer of at least 64 bytes");
    return NNHIP_E_INVALID;
  }
  static const char* names[] = {"NNHIP_EDGE_LDS", "NNHIP_EDGE_SMALL_ATOMS", "NNHIP_EDGE_WPR", "NNHIP_FORCE_DIRECT_MOL", "NNHIP_FORCE_FWD_MOL", "NNHIP_GRAPH_MOL",
                                "NNHIP_GRAPH_SMALL_ATOMS", "NNHIP_HEAD_OUT_MOL", "NNHIP_MLP_REGW", "NNHIP_MLP_REGW_SINGLE", "NNHIP_MLP_SPLIT",
                                "NNHIP_MLP_WIDE_TILES", "NNHIP_MOL_KERNELS_MIN", "NNHIP_MSG_BWD_MOL",
                                "NNHIP_WGRAD_FORM", "NNHIP_WGRAD_RPC"};
  int small_atoms, mol_min, wpr[4], mol_forms;
  edge_config(&small_atoms, &mol_min, wpr, &mol_forms);
  const int forms = nnhip_mlp_forms();
  const char* graph_mol = getenv("NNHIP_GRAPH_MOL");
  size_t o = 0;
  auto put = [&](const char* fmt, auto... a) {
    if (o < n) {
      const int k = snprintf(buf + o, n - o, fmt, a...);
      o += k > 0 ? (size_t)k : 0;
    }
  };
  put("{\"version\": %d, \"tooling_build\": %d, \"split_f16_products\": %d, ", nnhip_version(), nnhip_build_flags(), forms & 1);
  put("\"neighbor_list\": {\"single_launch_max_atoms\": %d, \"per_molecule_kernels\": %d, \"cell_list\": \"one periodic molecule (host choice)\"}, ",
      nnhip_graph_small_max_atoms(), (graph_mol && atoi(graph_mol) == 0) ? 0 : 1);
  put("\"edge_rows\": {\"waves_per_row\": {\"msg_fwd\": %d, \"force_fwd\": %d, \"force_bwd\": %d, \"msg_bwd\": %d}, \"four_waves_per_row_up_to_atoms\": %d}, ",
      wpr[0], wpr[1], wpr[2], wpr[3], small_atoms);
  put("\"molecule_forms\": {\"max_atoms\": %d, \"edge_kernels_from_molecules\": %d, \"force_fwd\": %d, \"msg_bwd\": %d, \"force_direct\": %d, \"head_out\": %d}, ",
      NNHIP_MOL_STAGE_MAX, mol_min, mol_forms & 1, (mol_forms >> 1) & 1, (mol_forms >> 2) & 1, (mol_forms >> 3) & 1);
  put("\"edge_mlp\": {\"row_local_up_to_tiles\": %d, \"one_pass_adjoint\": %d, \"one_pass_forward\": %d, \"one_pass_single_adjoint\": %d, "
      "\"one_pass_single_forward\": %d}, ",
      mlp_wide_max_tiles_silu(), (forms >> 1) & 1, (forms >> 2) & 1, (forms >> 3) & 1, (forms >> 4) & 1);
  put("\"radial_table_intervals\": %d, \"env\": {", FT_G);
  bool first = true;
  for (const char* nm : names) {
    const char* v = getenv(nm);
    if (!v) continue;
    // (the value goes into a JSON string: keep [A-Za-z0-9_.,+-] and blank out everything else -- a quote or a backslash in a switch
    // must not cost the caller its whole bench line)
    char clean[25];
    int k = 0;
    for (; k < 24 && v[k]; ++k) {
      const char ch = v[k];
      const bool ok = (ch >= '0' && ch <= '9') || (ch >= 'a' && ch <= 'z') || (ch >= 'A' && ch <= 'Z') || ch == '_' || ch == '.' ||
                      ch == ',' || ch == '+' || ch == '-';
      clean[k] = ok ? ch : '?';
    }
    clean[k] = 0;
    put("%s\"%s\": \"%s\"", first ? "" : ", ", nm, clean);
    first = false;
  }
  put("%s", "}}");
  if (o >= n) {
    nnhip_set_error("nnhip_config: buffer of %zu bytes is too small", n);
    return NNHIP_E_INVALID;
  }
  return NNHIP_OK;
}

// ---- timers ------------------------------------------------------------------------------------------
struct TimerRec {
  int cls;
  hipEvent_t e0, e1;
};
// (host threads driving different streams may time concurrently: the switch is atomic, the lists are under one mutex)
static std::atomic<bool> g_timers_on{false};
static std::atomic<uint32_t> g_timer_mask{0xffffffffu};   // classes that record events (nnhip_timers_enable)
static std::mutex g_timer_mu;
static std::vector<TimerRec> g_pending;
static std::vector<hipEvent_t> g_event_pool;
static double g_ms[NNHIP_N_TIMER_CLASSES];
static int64_t g_cnt[NNHIP_N_TIMER_CLASSES];

static hipEvent_t get_event() {
  {
    std::lock_guard<std::mutex> lk(g_timer_mu);
    if (!g_event_pool.empty()) {
      hipEvent_t e = g_event_pool.back();
      g_event_pool.pop_back();
      return e;
    }
  }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
ScopedTimer::ScopedTimer(int c, hipStream_t st)
    : cls(c), s(st), e0(nullptr), on(g_timers_on.load() && ((g_timer_mask.load() >> c) & 1u)) {
  if (on) {
    e0 = get_event();
    if (e0) (void)hipEventRecord(e0, s); else on = false;
  }
}
ScopedTimer::~ScopedTimer() {
  if (!on) return;
  hipEvent_t e1 = get_event();
  if (!e1) return;
  (void)hipEventRecord(e1, s);
  std::lock_guard<std::mutex> lk(g_timer_mu);
  g_pending.push_back({cls, e0, e1});
}
// on: 0 = off, 1 = every class, otherwise bit (k + 1) selects class k -- only the selected classes put events into the stream, so
// a class can be timed without the two events per launch of all the others (which delay every kernel boundary a little)
extern "C" int nnhip_timers_enable(int32_t on) {
  g_timers_on = on != 0;
  g_timer_mask = (on == 0 || on == 1) ? 0xffffffffu : ((uint32_t)on >> 1);
  return NNHIP_OK;
}
extern "C" int nnhip_timers_read(double* ms, int64_t* cnt, int32_t reset) {
  std::lock_guard<std::mutex> lk(g_timer_mu);
  for (auto& r : g_pending) {
    float t = 0.f;
    HIP_TRY(hipEventSynchronize(r.e1));
    HIP_TRY(hipEventElapsedTime(&t, r.e0, r.e1));
    g_ms[r.cls] += t;
    g_cnt[r.cls] += 1;
    g_event_pool.push_back(r.e0);
    g_event_pool.push_back(r.e1);
  }
  g_pending.clear();
  for (int k = 0; k < NNHIP_N_TIMER_CLASSES; ++k) {
    if (ms) ms[k] = g_ms[k];
    if (cnt) cnt[k] = g_cnt[k];
    if (reset) {
      g_ms[k] = 0;
      g_cnt[k] = 0;
    }
  }
  return NNHIP_OK;
}

// ---- workspace ---------------------------------------------------------------------------------------
struct WsInternal {
  nnhip_ws_layout pub;
  size_t prep;                     // PrepLayout block inside the workspace (fallback when none is passed)
  size_t xhat[NNHIP_MAX_LAYERS];   // [N][F] normalised atom_node of each layer (layer_norm=True only)
  size_t rstd[NNHIP_MAX_LAYERS];   // [N]    1/sigma of each row
  size_t g_h12;                    // [E][2F] adjoint scratch (g_phi -> g_h)
  size_t g_msg;                    // [E][F]
  size_t g_m;                      // [N][F]
  size_t g_e;                      // [N][F] head adjoint scratch (g_e2 then g_e1)
  size_t g_f2;                     // [N][3][F] second g_f buffer (ping-pong)
  size_t gf_mid;                   // [N][3][F] dE/d f_out of the layer after the update adjoint
  size_t g_d;                      // [E][4]
  size_t atom_energy;              // [N]
};
struct PrepLayout {
  size_t wT[NNHIP_MAX_LAYERS][7];  // transposed weights: node0, node2, eq1_0, eq1_2, eq2_0, eq2_2, update
  size_t headT[2];                 // head0^T, head2^T
  size_t hn_tab, m_tab;            // [128][F] message_nodepart of layer 0 evaluated on the embedding rows (per element)
  size_t ftab[NNHIP_MAX_LAYERS];   // radial-filter tables of each layer: planes T | S | D of [FT_ROWS][F] (graph.hip:filter_table_kernel)
  // split-f16 images (node128s.hip) of every [128][128] weight and its transpose (IMG_* of common.h), and of the head's
  size_t img[NNHIP_MAX_LAYERS][IMG_PER_LAYER];
  size_t img_head[IMG_HEAD_COUNT];
  size_t snap;                     // the parameter values the block was last checked against (nnhip_prepare_check), fp32 words
  size_t changes;                  // int32: bumped by nnhip_prepare_check_counter when a parameter differs from the snapshot; reset by nnhip_prepare
  size_t total;
};
static size_t prep_bytes(int L);
// every parameter tensor of a model, in a fixed order, with its size in floats (the snapshot layout of nnhip_prepare_check)
#define PREP_MAX_PARAMS (2 + 12 * NNHIP_MAX_LAYERS + 8)
static int model_params(const nnhip_model* m, const float** ptr, int* count) {
  const int nb = m->n_basis;
  int c = 0;
  auto add = [&](const float* p, int n) {
    ptr[c] = p;
    count[c++] = p ? n : 0;
  };
  add(m->node_embedding, NNHIP_N_ELEMENTS * NF);
  add(m->frequencies, nb);
  for (int l = 0; l < m->n_layers; ++l) {
    const nnhip_layer_params& lp = m->layer[l];
    add(lp.node0_w, NF * NF), add(lp.node0_b, NF), add(lp.node2_w, NF * NF), add(lp.node2_b, NF);
    add(lp.edge_w, NF * nb);
    add(lp.eq1_0_w, NF * NF), add(lp.eq1_2_w, NF * NF), add(lp.eq2_0_w, NF * NF), add(lp.eq2_2_w, NF * NF);
    add(lp.update_w, NF * NF), add(lp.ln_w, NF), add(lp.ln_b, NF);
  }
  add(m->head0_w, NF * NF), add(m->head0_b, NF), add(m->head2_w, NF * NF), add(m->head2_b, NF);
  add(m->head4_w, NF), add(m->head4_b, 1), add(m->scale, NNHIP_N_ELEMENTS), add(m->shift, NNHIP_N_ELEMENTS);
  return c;
}
static size_t snapshot_floats(int L) {   // capacity for the largest basis
  return (size_t)NNHIP_N_ELEMENTS * NF + NNHIP_MAX_NB + (size_t)L * (7 * NF * NF + 4 * NF + NF * NNHIP_MAX_NB) + 2 * NF * NF + 3 * NF +
         1 + 2 * NNHIP_N_ELEMENTS;
}

static size_t carve(size_t& off, size_t bytes) {
  const size_t o = off;
  off += (bytes + 255) & ~(size_t)255;
  return o;
}

// Everything that depends on the parameters only -- transposed weights for the reverse sweep, the radial-filter tables,
// layer 0's message_nodepart per element -- lives in one "prepared" block (offsets relative to its base) that
// nnhip_prepare fills.  Callers that do not pass one get it rebuilt inside the workspace on every call.
static void make_prep_layout(int L, PrepLayout& q) {
  memset(&q, 0, sizeof(q));
  size_t off = 0;
  for (int l = 0; l < L; ++l) {
    for (int k = 0; k < 7; ++k) q.wT[l][k] = carve(off, NF * NF * 4);
    q.ftab[l] = carve(off, 3 * FT_PLANE * 4);
  }
  q.hn_tab = carve(off, (size_t)128 * NF * 4);   // message_nodepart of layer 0 per element (119 rows, padded)
  q.m_tab = carve(off, (size_t)128 * NF * 4);
  q.headT[0] = carve(off, NF * NF * 4);
  q.headT[1] = carve(off, NF * NF * 4);
  for (int l = 0; l < L; ++l)
    for (int k = 0; k < IMG_PER_LAYER; ++k) q.img[l][k] = carve(off, WIMG_BYTES);
  for (int k = 0; k < IMG_HEAD_COUNT; ++k) q.img_head[k] = carve(off, WIMG_BYTES);
  q.snap = carve(off, snapshot_floats(L) * 4);
  q.changes = carve(off, 4);
  q.total = off;
}
static size_t prep_bytes(int L) {
  PrepLayout q;
  make_prep_layout(L, q);
  return q.total;
}

static void make_layout(int N, int E, int B, int L, WsInternal& w) {
  (void)B;
  memset(&w, 0, sizeof(w));
  size_t off = 0;
  const size_t nf = (size_t)N * NF * 4;
  const size_t ef = (size_t)((E + 1) / 2) * NF * 4;   // msg / h / phi / g_phi / g_msg live once per undirected pair
  const size_t hf = (size_t)(((E + 1) / 2 + 31) / 32 * 32) * NF * 4;   // hidden tiles: whole 32-row tiles (fragment order)
  w.pub.a0 = carve(off, nf);
  for (int l = 0; l < L; ++l) {
    w.pub.m[l] = carve(off, nf);
    w.pub.hn[l] = carve(off, nf);
    w.pub.msg[l] = carve(off, ef);
    w.pub.h12[l] = carve(off, 2 * hf);
    w.pub.phi1[l] = carve(off, ef);
    w.pub.phi2[l] = carve(off, ef);
    w.pub.a_mid[l] = carve(off, nf);
    w.pub.a_out[l] = carve(off, nf);
    w.pub.f_out[l] = carve(off, 3 * nf);
    w.pub.q[l] = carve(off, 3 * nf);
  }
  for (int l = 0; l < L; ++l) {
    w.xhat[l] = carve(off, nf);
    w.rstd[l] = carve(off, (size_t)N * 4);
  }
  w.pub.e1 = carve(off, nf);
  w.pub.e2 = carve(off, nf);
  w.pub.g_x = carve(off, (size_t)L * E * 4);
  w.pub.g_u = carve(off, (size_t)L * E * 16);
  w.pub.g_a = carve(off, nf);
  w.pub.g_f = carve(off, 3 * nf);
  w.g_f2 = carve(off, 3 * nf);
  w.gf_mid = carve(off, 3 * nf);
  w.g_h12 = carve(off, 2 * ef);
  w.g_msg = carve(off, ef);
  w.g_m = carve(off, nf);
  w.g_e = carve(off, nf);
  w.g_d = carve(off, (size_t)E * 16);
  w.atom_energy = carve(off, (size_t)N * 4);
  w.prep = carve(off, prep_bytes(L));   // used when the caller passes no prepared block
  w.pub.total = off;
}

extern "C" size_t nnhip_workspace_bytes(int32_t N, int32_t E, int32_t B, int32_t L) {
  if (N < 0 || E < 0 || L < 1 || L > NNHIP_MAX_LAYERS) return 0;
  WsInternal w;
  make_layout(N, E, B, L, w);
  return w.pub.total;
}

extern "C" int nnhip_workspace_layout(int32_t N, int32_t E, int32_t B, int32_t L, nnhip_ws_layout* out) {
  if (!out || N < 0 || E < 0 || L < 1 || L > NNHIP_MAX_LAYERS) {
    nnhip_set_error("nnhip_workspace_layout: bad arguments");
    return NNHIP_E_INVALID;
  }
  WsInternal w;
  make_layout(N, E, B, L, w);
  *out = w.pub;
  return NNHIP_OK;
}

#define TRY(x)            \
  do {                    \
    int _r = (x);         \
    if (_r) return _r;    \
  } while (0)

// ---- parameter-only preparation ---------------------------------------------------------------------------
static int run_prepare(const nnhip_model* model, const PrepLayout& pq, char* pbase, hipStream_t s) {
  const int L = model->n_layers;
  auto Q = [&](size_t off) { return (float*)(pbase + off); };
  // transposed weights for the reverse sweep
  {
    const float* src[40];
    float* dst[40];
    int c = 0;
    for (int l = 0; l < L; ++l) {
      const nnhip_layer_params& lp = model->layer[l];
      const float* ws_[7] = {lp.node0_w, lp.node2_w, lp.eq1_0_w, lp.eq1_2_w, lp.eq2_0_w, lp.eq2_2_w, lp.update_w};
      for (int k = 0; k < 7; ++k) {
        if (c == 40) {
          TRY(launch_transposes(src, dst, c, s));
          c = 0;
        }
        src[c] = ws_[k];
        dst[c] = Q(pq.wT[l][k]);
        ++c;
      }
    }
    if (c + 2 > 40) {
      TRY(launch_transposes(src, dst, c, s));
      c = 0;
    }
    src[c] = model->head0_w;
    dst[c++] = Q(pq.headT[0]);
    src[c] = model->head2_w;
    dst[c++] = Q(pq.headT[1]);
    TRY(launch_transposes(src, dst, c, s));
  }
  // split-f16 images of the node-level weights (after the transposes above: the reverse sweep's images are made from them)
  if (split_products_enabled() && model->activation == NNHIP_ACT_SILU) {
    const float* src[IMG_PER_LAYER * NNHIP_MAX_LAYERS + IMG_HEAD_COUNT];
    char* dst[IMG_PER_LAYER * NNHIP_MAX_LAYERS + IMG_HEAD_COUNT];
    int c = 0;
    for (int l = 0; l < L; ++l) {
      const nnhip_layer_params& lp = model->layer[l];
      const float* ws_[IMG_PER_LAYER];
      ws_[IMG_UPDATE] = lp.update_w;
      ws_[IMG_NODE0] = lp.node0_w;
      ws_[IMG_NODE2] = lp.node2_w;
      ws_[IMG_UPDATE_T] = Q(pq.wT[l][6]);
      ws_[IMG_NODE0_T] = Q(pq.wT[l][0]);
      ws_[IMG_NODE2_T] = Q(pq.wT[l][1]);
      ws_[IMG_EQ1_0] = lp.eq1_0_w;
      ws_[IMG_EQ1_2] = lp.eq1_2_w;
      ws_[IMG_EQ2_0] = lp.eq2_0_w;
      ws_[IMG_EQ2_2] = lp.eq2_2_w;
      ws_[IMG_EQ1_0_T] = Q(pq.wT[l][2]);
      ws_[IMG_EQ1_2_T] = Q(pq.wT[l][3]);
      ws_[IMG_EQ2_0_T] = Q(pq.wT[l][4]);
      ws_[IMG_EQ2_2_T] = Q(pq.wT[l][5]);
      for (int k = 0; k < IMG_PER_LAYER; ++k) {
        src[c] = ws_[k];
        dst[c++] = pbase + pq.img[l][k];
      }
    }
    const float* hs_[IMG_HEAD_COUNT] = {model->head0_w, model->head2_w, Q(pq.headT[0]), Q(pq.headT[1])};
    for (int k = 0; k < IMG_HEAD_COUNT; ++k) {
      src[c] = hs_[k];
      dst[c++] = pbase + pq.img_head[k];
    }
    TRY(launch_weight_images(src, dst, c, s));
  }
  // radial-filter tables, one per layer
  {
    const float* ew[NNHIP_MAX_LAYERS];
    float* tb[NNHIP_MAX_LAYERS];
    for (int l = 0; l < L; ++l) {
      ew[l] = model->layer[l].edge_w;
      tb[l] = Q(pq.ftab[l]);
    }
    TRY(launch_filter_tables(ew, tb, L, model->frequencies, model->n_basis, model->envelope, s));
  }
  // The first message_nodepart acts on Embedding[z]: evaluate it once per element (the 119 embedding rows); the atoms'
  // rows are looked up (embed_kernel) instead of pushing N identical-by-element rows through the MLP.  (hn of layer 0 is
  // not kept per atom: its adjoint is never needed, the embedding does not depend on the positions.)
  const nnhip_layer_params& l0 = model->layer[0];
  TRY(launch_mlp(MODE_FWD, false, {model->node_embedding, l0.node0_w, l0.node2_w, Q(pq.hn_tab), Q(pq.m_tab), NNHIP_N_ELEMENTS,
                                   NF, NF, NF, l0.node0_b, l0.node2_b, model->activation}, s));
  return NNHIP_OK;
}

static int check_model(const nnhip_model* model, const char* who) {
  if (!model) {
    nnhip_set_error("%s: bad arguments", who);
    return NNHIP_E_INVALID;
  }
  if (model->n_features != NF || model->n_basis < 1 || model->n_basis > NNHIP_MAX_NB || model->n_layers < 1 ||
      model->n_layers > NNHIP_MAX_LAYERS) {
    nnhip_set_error("%s: n_features=%d n_basis=%d n_layers=%d unsupported (built for %d / 1..%d / 1..%d)", who,
                    model->n_features, model->n_basis, model->n_layers, NF, NNHIP_MAX_NB, NNHIP_MAX_LAYERS);
    return NNHIP_E_UNSUPPORTED;
  }
  return NNHIP_OK;
}

extern "C" size_t nnhip_prepared_bytes(int32_t L) {
  if (L < 1 || L > NNHIP_MAX_LAYERS) return 0;
  return prep_bytes(L);
}


// Did a parameter change since this block was last checked?  Exact (bitwise) comparison of every parameter tensor with the
// snapshot kept inside the block; the snapshot is brought up to date in the same pass and `bit` is OR-ed into *status when
// anything differed.  NewtonNet.forward runs it ahead of the edge-count read-back and reads the answer with the count: the
// block is refilled (nnhip_prepare) only when the answer says so -- nothing is keyed on tensor identity or version counters,
// which in-place writers outside torch (nnhip_clip_adam) and recycled allocations defeat.
struct ParamTable {
  int n;
  const float* src[PREP_MAX_PARAMS];
  int count[PREP_MAX_PARAMS];
  int off[PREP_MAX_PARAMS];
};
// one workgroup per 1024 consecutive words of one tensor, four words per thread (all loads independent): ~400 workgroups, ~3 us
#define PARAM_CHECK_MAX_CHUNKS ((NNHIP_N_ELEMENTS * NF + 1023) / 1024 > NF * NF / 1024 ? (NNHIP_N_ELEMENTS * NF + 1023) / 1024 : NF * NF / 1024)
// UPDATE = false: compare only -- OR `bit` into *status when a word differs (the snapshot is left alone, so a change stays
// visible until nnhip_prepare refills the block: a caller that fails between the check and the refill cannot lose it);
// UPDATE = true: copy the parameters into the snapshot (the tail of nnhip_prepare).
// The blocks of column blockIdx.x == t.n (when the grid has it) clear the neighbor list's status word, molecule extents and
// row_ptr instead (graph.hip:graph_init_kernel's job, riding in this launch: the deferred step's one launch less).
struct GraphInit {
  int* status; int* mol_ptr; int n_mol1; int* row_ptr; int n_atoms1;
};
// ... and the columns behind that one do edge.hip:embed_kernel's job (atom_node = Embedding[z] and the first layer's m from its
// per-element table: a gather that depends on nothing but z and the prepared block -- if this very launch finds the block stale,
// the step is repeated anyway): the deferred step's one launch less.
struct EmbedJob {
  const int64_t* z; const float* table; const float* m_table; int n_atoms; float* a0; float* m0;
};
template <bool UPDATE>
__global__ void __launch_bounds__(256) param_check_kernel(ParamTable t, uint32_t* __restrict__ snap, int* __restrict__ status, int bit,
                                                          GraphInit gi = GraphInit{nullptr, nullptr, 0, nullptr, 0},
                                                          EmbedJob ej = EmbedJob{nullptr, nullptr, nullptr, 0, nullptr, nullptr}) {
  const int k = blockIdx.x;
  const int first_embed = t.n + (gi.status ? 1 : 0);
  if (k >= first_embed) {
    const size_t e = ((size_t)(k - first_embed) * gridDim.y + blockIdx.y) * 256 + threadIdx.x;   // one float4 each
    if (e >= (size_t)ej.n_atoms * (NF / 4)) return;
    const int i = (int)(e / (NF / 4)), c = (int)(e % (NF / 4));
    const size_t zi = (size_t)clamp_species(ej.z[i]);
    reinterpret_cast<float4*>(ej.a0)[e] = reinterpret_cast<const float4*>(ej.table + zi * NF)[c];
    reinterpret_cast<float4*>(ej.m0)[e] = reinterpret_cast<const float4*>(ej.m_table + zi * NF)[c];
    return;
  }
  if (k == t.n) {
    const int n_init = gi.n_mol1 > gi.n_atoms1 ? gi.n_mol1 : gi.n_atoms1;
    for (int i = blockIdx.y * 256 + threadIdx.x; i < n_init; i += 256 * gridDim.y) {
      if (i == 0) gi.status[0] = 0;
      if (i < gi.n_mol1) gi.mol_ptr[i] = 0;
      if (i < gi.n_atoms1) gi.row_ptr[i] = 0;
    }
    return;
  }
  const int n = t.count[k];
  const int e0 = (blockIdx.y * 256 + threadIdx.x) * 4;
  if (blockIdx.y * 1024 >= n) return;   // (block-uniform)
  const uint32_t* __restrict__ src = reinterpret_cast<const uint32_t*>(t.src[k]);
  uint32_t* __restrict__ dst = snap + t.off[k];
  uint32_t v[4], w[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    v[q] = e0 + q < n ? src[e0 + q] : 0u;
    w[q] = e0 + q < n ? dst[e0 + q] : 0u;
  }
  int differs = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (v[q] != w[q]) {
      if (UPDATE) dst[e0 + q] = v[q];
      differs = 1;
    }
  // compare: bit `bit` into *status when bit != 0, else one more count in *status (nnhip_prepare_check_counter)
  if (!UPDATE && __syncthreads_or(differs) && threadIdx.x == 0) {
    if (bit) atomicOr(status, bit); else atomicAdd(status, 1);
  }
  if (UPDATE && status && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *status = 0;   // (nnhip_prepare resets the counter)
}
static int make_param_table(const nnhip_model* model, ParamTable& t) {
  memset(&t, 0, sizeof(t));
  const float* ptr[PREP_MAX_PARAMS];
  int count[PREP_MAX_PARAMS];
  const int n = model_params(model, ptr, count);
  int off = 0;
  for (int k = 0; k < n; ++k) {
    if (count[k] == 0) continue;
    t.src[t.n] = ptr[k];
    t.count[t.n] = count[k];
    t.off[t.n] = off;
    off += count[k];
    ++t.n;
  }
  if ((size_t)off > snapshot_floats(model->n_layers)) {
    nnhip_set_error("nnhip_prepare: snapshot overflow");
    return NNHIP_E_INVALID;
  }
  return NNHIP_OK;
}

extern "C" int nnhip_prepare_check(const nnhip_model* model, void* prepared, size_t prepared_bytes, int32_t* status, int32_t bit,
                                   void* stream_) {
  TRY(check_model(model, "nnhip_prepare_check"));
  PrepLayout pq;
  make_prep_layout(model->n_layers, pq);
  if (!prepared || !status || prepared_bytes < pq.total || ((uintptr_t)prepared & 255) != 0) {
    nnhip_set_error("nnhip_prepare_check: block of %zu bytes (need %zu, 256-byte aligned)", prepared_bytes, pq.total);
    return NNHIP_E_WORKSPACE;
  }
  ParamTable t;
  TRY(make_param_table(model, t));
  param_check_kernel<false><<<dim3(t.n, PARAM_CHECK_MAX_CHUNKS), 256, 0, (hipStream_t)stream_>>>(
      t, reinterpret_cast<uint32_t*>((char*)prepared + pq.snap), status, bit);
  LAUNCH_CHECK();
  return NNHIP_OK;
}

// Fill a prepared block AND take the snapshot of the parameters it was filled from (what nnhip_prepare_check compares with).
extern "C" int nnhip_prepare(const nnhip_model* model, void* prepared, size_t prepared_bytes, void* stream_) {
  TRY(check_model(model, "nnhip_prepare"));
  PrepLayout pq;
  make_prep_layout(model->n_layers, pq);
  if (!prepared || prepared_bytes < pq.total || ((uintptr_t)prepared & 255) != 0) {
    nnhip_set_error("nnhip_prepare: block of %zu bytes (need %zu, 256-byte aligned)", prepared_bytes, pq.total);
    return NNHIP_E_WORKSPACE;
  }
  TRY(run_prepare(model, pq, (char*)prepared, (hipStream_t)stream_));
  ParamTable t;
  TRY(make_param_table(model, t));
  param_check_kernel<true><<<dim3(t.n, PARAM_CHECK_MAX_CHUNKS), 256, 0, (hipStream_t)stream_>>>(
      t, reinterpret_cast<uint32_t*>((char*)prepared + pq.snap), reinterpret_cast<int*>((char*)prepared + pq.changes), 0);
  LAUNCH_CHECK();
  return NNHIP_OK;
}

// nnhip_prepare_check reporting through a counter INSIDE the block instead of a caller's status word: the counter is zero after
// nnhip_prepare and goes up whenever a check finds a parameter that differs from the snapshot (it keeps going up until the block
// is refilled).  No word of the caller has to be initialised for it -- the deferred step runs it first and lets its last
// neighbor-list kernel hand the counter to the host with the edge count.  *counter_out receives the counter's device address.
static int prepare_check_counter_impl(const nnhip_model* model, void* prepared, size_t prepared_bytes, const int32_t** counter_out,
                                      void* stream_, const GraphInit& gi,
                                      const EmbedJob& ej = EmbedJob{nullptr, nullptr, nullptr, 0, nullptr, nullptr});
extern "C" int nnhip_prepare_check_counter(const nnhip_model* model, void* prepared, size_t prepared_bytes, const int32_t** counter_out,
                                           void* stream_) {
  return prepare_check_counter_impl(model, prepared, prepared_bytes, counter_out, stream_, GraphInit{nullptr, nullptr, 0, nullptr, 0});
}
static int prepare_check_counter_impl(const nnhip_model* model, void* prepared, size_t prepared_bytes, const int32_t** counter_out,
                                      void* stream_, const GraphInit& gi, const EmbedJob& ej) {
  TRY(check_model(model, "nnhip_prepare_check_counter"));
  PrepLayout pq;
  make_prep_layout(model->n_layers, pq);
  if (!prepared || prepared_bytes < pq.total || ((uintptr_t)prepared & 255) != 0) {
    nnhip_set_error("nnhip_prepare_check_counter: block of %zu bytes (need %zu, 256-byte aligned)", prepared_bytes, pq.total);
    return NNHIP_E_WORKSPACE;
  }
  ParamTable t;
  TRY(make_param_table(model, t));
  int* counter = reinterpret_cast<int*>((char*)prepared + pq.changes);
  const int embed_cols = ej.a0 ? (int)(((size_t)ej.n_atoms * (NF / 4) + 256 * PARAM_CHECK_MAX_CHUNKS - 1) / (256 * PARAM_CHECK_MAX_CHUNKS)) : 0;
  param_check_kernel<false><<<dim3(t.n + (gi.status ? 1 : 0) + embed_cols, PARAM_CHECK_MAX_CHUNKS), 256, 0, (hipStream_t)stream_>>>(
      t, reinterpret_cast<uint32_t*>((char*)prepared + pq.snap), counter, 0, gi, ej);
  LAUNCH_CHECK();
  if (counter_out) *counter_out = counter;
  return NNHIP_OK;
}

// ---- the hot path --------------------------------------------------------------------------------------
// n_pairs_dev == NULL: E is the edge count.  Otherwise (nnhip_energy_forces_dev) E is the CAPACITY the per-edge arrays and the
// workspace are sized for and *n_pairs_dev the true number of undirected pairs (= pair_ptr[N]): the row kernels walk row_ptr, the
// per-edge kernels cover the capacity (rows beyond the count hold nothing anybody reads), the pair-row kernels read the count.
static int energy_forces_impl(const nnhip_model* model, const int64_t* z, const float* pos, const float* cell,
                              const int32_t* mol_ptr, const int32_t* row_ptr, const int32_t* col,
                              const int32_t* rev, const int32_t* pid, const float* geo, const int32_t* xg,
                              const float* disp, int32_t N, int32_t E, int32_t B, void* workspace,
                              size_t workspace_bytes, float* energy, float* forces, float* virial,
                              float* atom_energy_out, float* atom_node_out, float* force_node_out,
                              const void* prepared, const int32_t* n_pairs_dev, const int32_t* pair_ptr, void* stream_,
                              bool mol_kernels = false, bool embedded = false) {
  hipStream_t s = (hipStream_t)stream_;
  if (!model || !energy || N < 0 || E < 0 || B < 0) {
    nnhip_set_error("nnhip_energy_forces: bad arguments");
    return NNHIP_E_INVALID;
  }
  if (model->n_features != NF || model->n_basis < 1 || model->n_basis > NNHIP_MAX_NB || model->n_layers < 1 ||
      model->n_layers > NNHIP_MAX_LAYERS) {
    nnhip_set_error("nnhip_energy_forces: n_features=%d n_basis=%d n_layers=%d unsupported (built for %d / 1..%d / 1..%d)",
                    model->n_features, model->n_basis, model->n_layers, NF, NNHIP_MAX_NB, NNHIP_MAX_LAYERS);
    return NNHIP_E_UNSUPPORTED;
  }
  const int L = model->n_layers;
  const int act = model->activation;
  // explicit candidate mask in the force kernels: only when act(0) != 0 (edge.hip:force_fwd_kernel)
  const int32_t* mask_xg = (act == NNHIP_ACT_SIGMOID || act == NNHIP_ACT_SOFTPLUS) ? xg : nullptr;
  if (act < NNHIP_ACT_SILU || act > NNHIP_ACT_SSP) {
    nnhip_set_error("nnhip_energy_forces: unknown activation id %d", act);
    return NNHIP_E_UNSUPPORTED;
  }
  for (int l = 0; l < L; ++l)
    if ((model->layer[l].ln_w == nullptr) != (model->layer[l].ln_b == nullptr)) {
      nnhip_set_error("nnhip_energy_forces: layer %d has only one of layer_norm.weight / .bias", l);
      return NNHIP_E_INVALID;
    }
  if (E & 1) {
    nnhip_set_error("nnhip_energy_forces: odd edge count %d (the edge set must be symmetric)", E);
    return NNHIP_E_INVALID;
  }
  const int P_ = E / 2;   // undirected pairs
  // (pair_ptr: the pair counts per row, when the caller has them -- the row kernels then split a row into the pairs it owns and
  // the others with two scalar loads)
  const size_t h2_off = (size_t)((P_ + 31) / 32 * 32) * NF;   // floats between the h1 and h2 regions of a layer
  WsInternal w;
  make_layout(N, E, B, L, w);
  if (workspace_bytes < w.pub.total || (!workspace && w.pub.total)) {
    nnhip_set_error("nnhip_energy_forces: workspace %zu < required %zu bytes", workspace_bytes, w.pub.total);
    return NNHIP_E_WORKSPACE;
  }
  if (((uintptr_t)workspace & 255) != 0) {
    nnhip_set_error("nnhip_energy_forces: workspace must be 256-byte aligned");
    return NNHIP_E_INVALID;
  }
  if (N == 0) {
    if (B > 0) HIP_TRY(hipMemsetAsync(energy, 0, sizeof(float) * B, s));
    return NNHIP_OK;
  }
  char* ws = (char*)workspace;
  auto P = [&](size_t off) { return (float*)(ws + off); };
  const bool want_forces = forces != nullptr;

  // parameter-only data: use the caller's prepared block, or rebuild it in the workspace
  PrepLayout pq;
  make_prep_layout(L, pq);
  char* pbase = prepared ? (char*)prepared : ws + w.prep;
  auto Q = [&](size_t off) { return (float*)(pbase + off); };
  if (!prepared) TRY(run_prepare(model, pq, pbase, s));
  const bool split_nodes = split_products_enabled() && act == NNHIP_ACT_SILU;   // node128s.hip (images in the prepared block)
  // ------------------------------------------------------------------ forward sweep
  // The first message_nodepart acts on Embedding[z]: evaluate it once per element (the 119 embedding rows) and look the
  // atoms' rows up, instead of pushing N identical-by-element rows through the MLP.  (hn of layer 0 is not kept: its
  // adjoint is never needed, the embedding does not depend on the positions.)
  // (embedded: the caller's parameter-check launch has done this gather already, nnhip_forward_dev)
  if (!embedded) TRY(launch_embed(z, model->node_embedding, Q(pq.m_tab), N, P(w.pub.a0), P(w.pub.m[0]), s));
  const float* a_in = P(w.pub.a0);
  const float* f_in = nullptr;  // force_node == 0 entering the first layer (newtonnet.py:143)
  // the last layer writes atom_node / force_node straight into the caller's output arrays when they are given
  auto A_OUT = [&](int l) { return (l == L - 1 && atom_node_out) ? atom_node_out : P(w.pub.a_out[l]); };
  auto F_OUT = [&](int l) { return (l == L - 1 && force_node_out) ? force_node_out : P(w.pub.f_out[l]); };
  for (int l = 0; l < L; ++l) {
    const nnhip_layer_params& lp = model->layer[l];
    const bool has_f = l > 0;
    // message_nodepart (hn = a W0^T + b0 ; m = silu(hn) W2^T + b2) was produced by the fused node kernel that closed the
    // previous layer (by the per-element table for l = 0)
    // messages + invariant update
    {
      TRY(launch_msg_fwd(P(w.pub.m[l]), xg, Q(pq.ftab[l]), row_ptr, col, pid, a_in, P(w.pub.msg[l]), P(w.pub.a_mid[l]), N, s));
      // equiv_message{1,2}: h12 = msg [V1_0 ; V2_0]^T ; phi_k = silu(h_k) V_k2^T   (layer 0: phi2 multiplies force_node == 0)
      if (E > 0) {  // fused Linear -> SiLU -> Linear per MLP; h1 | h2 are kept interleaved in h12[E][2F] for the adjoint
        float* h12 = P(w.pub.h12[l]);   // h1 | h2: two pad32(P) x F regions, private to the MLP kernels (fragment order)
        MlpArgs m1 = {P(w.pub.msg[l]), lp.eq1_0_w, lp.eq1_2_w, h12, P(w.pub.phi1[l]), P_, NF, NF, NF};
        MlpArgs m2 = {P(w.pub.msg[l]), lp.eq2_0_w, lp.eq2_2_w, h12 + h2_off, P(w.pub.phi2[l]), P_, NF, NF, NF};
        m1.h_frag = m2.h_frag = 1;
        m1.act = m2.act = act;
        m1.M_dev = m2.M_dev = n_pairs_dev;
        if (split_nodes) {   // (used by the row-local form only: small pair counts)
          m1.W1_img = pbase + pq.img[l][IMG_EQ1_0];
          m1.W2_img = pbase + pq.img[l][IMG_EQ1_2];
          m2.W1_img = pbase + pq.img[l][IMG_EQ2_0];
          m2.W2_img = pbase + pq.img[l][IMG_EQ2_2];
        }
        if (has_f)
          TRY(launch_mlp_pair(MODE_FWD, m1, false, m2, false, s));
        else
          TRY(launch_mlp(MODE_FWD, false, m1, s));
      }
      TRY(launch_force_fwd(has_f, P(w.pub.phi1[l]), P(w.pub.phi2[l]), geo, row_ptr, col, pid, f_in, F_OUT(l), N, mask_xg, s, pair_ptr,
                           mol_kernels ? mol_ptr : nullptr, B));
    }
    // equiv_update + energy update + the next layer's message_nodepart: one row-local launch (node128.hip)
    {
      NodeFwdArgs na;
      memset(&na, 0, sizeof(na));
      na.f = F_OUT(l);
      na.a_mid = P(w.pub.a_mid[l]);
      na.Wu = lp.update_w;
      na.q = P(w.pub.q[l]);
      na.a_out = A_OUT(l);
      if (l + 1 < L && !lp.ln_w) {
        const nnhip_layer_params& nx = model->layer[l + 1];
        na.W0 = nx.node0_w;
        na.b0 = nx.node0_b;
        na.W2 = nx.node2_w;
        na.b2 = nx.node2_b;
        na.hn = P(w.pub.hn[l + 1]);
        na.m = P(w.pub.m[l + 1]);
      } else if (l + 1 == L && !lp.ln_w) {
        // after the last layer the same slot runs the first two linears of the energy head (output.py:90-95):
        // e1 = a W0^T + b0 ; e2 = silu(e1) W2^T + b2
        na.W0 = model->head0_w;
        na.b0 = model->head0_b;
        na.W2 = model->head2_w;
        na.b2 = model->head2_b;
        na.hn = P(w.pub.e1);
        na.m = P(w.pub.e2);
      }
      na.N = N;
      na.act = act;
      if (split_nodes) {
        NodeImages im;
        memset(&im, 0, sizeof(im));
        im.Wu = pbase + pq.img[l][IMG_UPDATE];
        if (na.W0) {
          im.W0 = l + 1 < L ? pbase + pq.img[l + 1][IMG_NODE0] : pbase + pq.img_head[IMG_HEAD0];
          im.W2 = l + 1 < L ? pbase + pq.img[l + 1][IMG_NODE2] : pbase + pq.img_head[IMG_HEAD2];
        }
        TRY(launch_node_fwd_split(na, im, s));
      } else {
        TRY(launch_node_fwd(na, s));
      }
    }
    if (lp.ln_w) {   // layer_norm=True (newtonnet.py:228-231): normalise in place, then the next message_nodepart unfused
      TRY(launch_layer_norm_fwd(A_OUT(l), lp.ln_w, lp.ln_b, N, P(w.xhat[l]), P(w.rstd[l]), s));
      if (l + 1 < L) {
        const nnhip_layer_params& nx = model->layer[l + 1];
        TRY(launch_mlp(MODE_FWD, false, {A_OUT(l), nx.node0_w, nx.node2_w, P(w.pub.hn[l + 1]), P(w.pub.m[l + 1]), N, NF, NF,
                                         NF, nx.node0_b, nx.node2_b, act}, s));
      }
    }
    a_in = A_OUT(l);
    f_in = F_OUT(l);
  }
  // energy head (fused into the last node launch unless that layer ends in a LayerNorm)
  if (model->layer[L - 1].ln_w)
    TRY(launch_mlp(MODE_FWD, false, {a_in, model->head0_w, model->head2_w, P(w.pub.e1), P(w.pub.e2), N, NF, NF, NF,
                                     model->head0_b, model->head2_b, act}, s));
  float* atom_energy = atom_energy_out ? atom_energy_out : P(w.atom_energy);
  TRY(launch_head_out(P(w.pub.e2), model->head4_w, model->head4_b, model->scale, model->shift, z, mol_ptr, N, B,
                      act, atom_energy, want_forces ? P(w.g_e) : nullptr, energy, s, mol_kernels));
  if (!want_forces) return NNHIP_OK;

  // ------------------------------------------------------------------ reverse sweep
  auto node_bwd = [&](const NodeBwdArgs& a, const NodeImages& im) {
    return split_nodes ? launch_node_bwd_split(a, im, s) : launch_node_bwd(a, s);
  };
  // head adjoint (g_e1 = (g_e2 H2) * silu'(e1); g_a = g_e1 H0) + update adjoint of the last layer
  // (gf = g_a * q + (g_a * f) W_u; dE/d force_node after the last layer is zero): one row-local launch
  {
    NodeBwdArgs nb;
    memset(&nb, 0, sizeof(nb));
    nb.g_top = P(w.g_e);
    nb.h_top = P(w.pub.e1);
    nb.W2T = Q(pq.headT[1]);
    nb.W0T = Q(pq.headT[0]);
    nb.g_a = P(w.pub.g_a);
    nb.acc_ga = 0;
    nb.f = F_OUT(L - 1);
    nb.q = P(w.pub.q[L - 1]);
    nb.G_f = nullptr;
    nb.WuT = Q(pq.wT[L - 1][6]);
    nb.gf = P(w.gf_mid);
    nb.N = N;
    nb.act = act;
    NodeImages bim;
    memset(&bim, 0, sizeof(bim));
    bim.W2T = pbase + pq.img_head[IMG_HEAD2_T];
    bim.W0T = pbase + pq.img_head[IMG_HEAD0_T];
    bim.WuT = pbase + pq.img[L - 1][IMG_UPDATE_T];
    const nnhip_layer_params& top = model->layer[L - 1];
    if (top.ln_w) {   // the LayerNorm adjoint sits between the head adjoint and the update adjoint: three launches
      NodeBwdArgs head = nb;
      head.WuT = nullptr;
      TRY(node_bwd(head, bim));
      TRY(launch_layer_norm_bwd(P(w.pub.g_a), top.ln_w, P(w.xhat[L - 1]), P(w.rstd[L - 1]), N, s));
      nb.W2T = nullptr;
    }
    TRY(node_bwd(nb, bim));
  }
  float* g_fbuf[2] = {P(w.pub.g_f), P(w.g_f2)};
  int pp = 0;
  for (int l = L - 1; l >= 0; --l) {
    const nnhip_layer_params& lp = model->layer[l];
    (void)lp;
    const bool has_f = l > 0;
    const float* f_prev = has_f ? F_OUT(l - 1) : nullptr;
    // force-message adjoint
    float* g_fin = g_fbuf[pp];
    {
      TRY(launch_force_bwd(has_f, P(w.gf_mid), P(w.pub.phi1[l]), P(w.pub.phi2[l]), geo, row_ptr, col, pid, f_prev,
                           P(w.g_h12), P(w.pub.g_u) + (size_t)l * E * 4, g_fin, N, mask_xg, s, pair_ptr, rev));
      if (E > 0) {
        // g_msg = ((g_phi1 V12) * silu'(h1)) V10 + ((g_phi2 V22) * silu'(h2)) V20, each term one fused launch
        float* gp = P(w.g_h12);   // [P][2F]: g_phi1 | g_phi2 written by force_bwd (pair space)
        float* h12 = P(w.pub.h12[l]);
        MlpArgs m1 = {gp, Q(pq.wT[l][3]), Q(pq.wT[l][2]), h12, P(w.g_msg), P_, 2 * NF, NF, NF};
        MlpArgs m2 = {gp + NF, Q(pq.wT[l][5]), Q(pq.wT[l][4]), h12 + h2_off, P(w.g_msg), P_, 2 * NF, NF, NF};
        m1.h_frag = m2.h_frag = 1;
        m1.act = m2.act = act;
        m1.M_dev = m2.M_dev = n_pairs_dev;
        if (split_nodes) {
          m1.W1_img = pbase + pq.img[l][IMG_EQ1_2_T];
          m1.W2_img = pbase + pq.img[l][IMG_EQ1_0_T];
          m2.W1_img = pbase + pq.img[l][IMG_EQ2_2_T];
          m2.W2_img = pbase + pq.img[l][IMG_EQ2_0_T];
        }
        if (has_f)
          TRY(launch_mlp_pair(MODE_BWD, m1, false, m2, true, s));
        else
          TRY(launch_mlp(MODE_BWD, false, m1, s));
      }
      // message adjoint -> g_m, g_x
      TRY(launch_msg_bwd(P(w.g_msg), P(w.pub.g_a), P(w.pub.m[l]), xg, Q(pq.ftab[l]), row_ptr, col, pid, P(w.g_m),
                         P(w.pub.g_x) + (size_t)l * E, N, l > 0, s, pair_ptr, mol_kernels ? mol_ptr : nullptr, B));
    }
    // message_nodepart adjoint of this layer (g_hn = (g_m W2) * silu'(hn); g_a += g_hn W0) + update adjoint of the
    // layer below (gf = G_f + g_a * q + (g_a * f) W_u): one row-local launch.  Nothing to do below the first layer: its
    // message_nodepart input is the embedding of z, which does not depend on the positions.
    if (l > 0) {
      NodeBwdArgs nb;
      memset(&nb, 0, sizeof(nb));
      nb.g_top = P(w.g_m);
      nb.h_top = P(w.pub.hn[l]);
      nb.W2T = Q(pq.wT[l][1]);
      nb.W0T = Q(pq.wT[l][0]);
      nb.g_a = P(w.pub.g_a);
      nb.acc_ga = 1;
      nb.f = F_OUT(l - 1);
      nb.q = P(w.pub.q[l - 1]);
      nb.G_f = g_fin;   // dE/d f_out of layer l-1, produced by force_bwd of layer l just above
      nb.WuT = Q(pq.wT[l - 1][6]);
      nb.gf = P(w.gf_mid);
      nb.N = N;
      nb.act = act;
      NodeImages bim;
      memset(&bim, 0, sizeof(bim));
      bim.W2T = pbase + pq.img[l][IMG_NODE2_T];
      bim.W0T = pbase + pq.img[l][IMG_NODE0_T];
      bim.WuT = pbase + pq.img[l - 1][IMG_UPDATE_T];
      const nnhip_layer_params& below = model->layer[l - 1];
      if (below.ln_w) {
        NodeBwdArgs mlp = nb;
        mlp.WuT = nullptr;
        TRY(node_bwd(mlp, bim));
        TRY(launch_layer_norm_bwd(P(w.pub.g_a), below.ln_w, P(w.xhat[l - 1]), P(w.rstd[l - 1]), N, s));
        nb.W2T = nullptr;
      }
      TRY(node_bwd(nb, bim));
    }
    pp ^= 1;
  }
  TRY(launch_geometry_bwd(P(w.pub.g_x), P(w.pub.g_u), geo, disp, pos, cell, row_ptr, col, rev, mol_ptr, N, E, B, L,
                          model->cutoff, P(w.g_d), forces, virial, s, mol_kernels));
  return NNHIP_OK;
}

extern "C" int nnhip_energy_forces(const nnhip_model* model, const int64_t* z, const float* pos, const float* cell,
                                   const int32_t* mol_ptr, const int32_t* row_ptr, const int32_t* col,
                                   const int32_t* rev, const int32_t* pid, const float* geo, const int32_t* xg,
                                   const float* disp, int32_t N, int32_t E, int32_t B, void* workspace,
                                   size_t workspace_bytes, float* energy, float* forces, float* virial,
                                   float* atom_energy_out, float* atom_node_out, float* force_node_out,
                                   const void* prepared, void* stream_) {
  return energy_forces_impl(model, z, pos, cell, mol_ptr, row_ptr, col, rev, pid, geo, xg, disp, N, E, B, workspace, workspace_bytes,
                            energy, forces, virial, atom_energy_out, atom_node_out, force_node_out, prepared, nullptr, nullptr,
                            stream_);
}
// nnhip_energy_forces for a caller that also has pair_ptr[N + 1] (nnhip_graph_count_pairs + nnhip_graph_pair_scan; what
// newtonnet_amd/hip.py:build_graph keeps): same step, same results.  flags bit 0: no molecule has more than NNHIP_MOL_STAGE_MAX
// atoms (the graph's status word has bit 8 clear): the molecule-resident edge kernels may run.
extern "C" int nnhip_energy_forces_pp(const nnhip_model* model, const int64_t* z, const float* pos, const float* cell,
                                      const int32_t* mol_ptr, const int32_t* row_ptr, const int32_t* col,
                                      const int32_t* rev, const int32_t* pid, const float* geo, const int32_t* xg,
                                      const float* disp, int32_t N, int32_t E, int32_t B, void* workspace,
                                      size_t workspace_bytes, float* energy, float* forces, float* virial,
                                      float* atom_energy_out, float* atom_node_out, float* force_node_out,
                                      const void* prepared, const int32_t* pair_ptr, int32_t flags, void* stream_) {
  return energy_forces_impl(model, z, pos, cell, mol_ptr, row_ptr, col, rev, pid, geo, xg, disp, N, E, B, workspace, workspace_bytes,
                            energy, forces, virial, atom_energy_out, atom_node_out, force_node_out, prepared, nullptr, pair_ptr,
                            stream_, (flags & 1) != 0);
}
// The same step queued BEFORE the host knows the edge count (NewtonNet.forward's steady state: no device->host round trip inside
// a step).  `capacity` (even, > 0) sizes the per-edge arrays (nnhip_graph_finish_dev) and the workspace
// (nnhip_workspace_bytes(N, capacity, ...)); n_pairs_dev = &pair_ptr[N].  When the count exceeded the capacity the graph arrives
// here emptied by nnhip_graph_finish_dev's guard: the step runs on zero edges, in bounds, and the host repeats it.
extern "C" int nnhip_energy_forces_dev(const nnhip_model* model, const int64_t* z, const float* pos, const float* cell,
                                       const int32_t* mol_ptr, const int32_t* row_ptr, const int32_t* col,
                                       const int32_t* rev, const int32_t* pid, const float* geo, const int32_t* xg,
                                       const float* disp, int32_t N, int32_t capacity, int32_t B, void* workspace,
                                       size_t workspace_bytes, float* energy, float* forces, float* virial,
                                       float* atom_energy_out, float* atom_node_out, float* force_node_out,
                                       const void* prepared, const int32_t* n_pairs_dev, void* stream_) {
  if (!n_pairs_dev || capacity < 2 || (capacity & 1)) {
    nnhip_set_error("nnhip_energy_forces_dev: needs n_pairs_dev and an even capacity > 0");
    return NNHIP_E_INVALID;
  }
  return energy_forces_impl(model, z, pos, cell, mol_ptr, row_ptr, col, rev, pid, geo, xg, disp, N, capacity, B, workspace,
                            workspace_bytes, energy, forces, virial, atom_energy_out, atom_node_out, force_node_out, prepared,
                            n_pairs_dev, n_pairs_dev - N, stream_);
}

// ---- the deferred step as one call -----------------------------------------------------------------------------
extern "C" int nnhip_step_layout_of(int32_t N, int32_t B, int32_t cap, nnhip_step_layout* out) {
  if (!out || N < 0 || B < 0 || cap < 2 || (cap & 1)) {
    nnhip_set_error("nnhip_step_layout_of: bad arguments");
    return NNHIP_E_INVALID;
  }
  memset(out, 0, sizeof(*out));
  const size_t n_scan = (size_t)(N + 1023) / 1024 + 1;
  size_t o = 0;
  auto take = [&](size_t n) {   // 16-byte granules: the per-edge arrays are read as int2 / float4
    const size_t at = o;
    o += (n + 3) & ~(size_t)3;
    return at;
  };
  out->mol_ptr = take((size_t)B + 1);
  out->row_ptr = take((size_t)N + 1 + 1 + n_scan);   // row_ptr [N + 1] | status word | the scan scratch behind it
  out->status = out->row_ptr + N + 1;                  // (adjacent to row_ptr[N]: ONE 8-byte copy brings count and status)
  out->pair_ptr = take((size_t)N + 1);
  out->pair_scan = take(n_scan);
  out->tail = take(2);
  out->mol_scratch = take(2 * (size_t)B + (size_t)B / 1024 + 4);   // nnhip_graph_mol_dev: molecule totals (+ their scan)
  out->xg = take(2 * (size_t)cap);
  out->col = take(cap);
  out->rev = take(cap);
  out->pid = take(cap);
  out->i32_count = o;
  o = 0;
  out->geo = take(4 * (size_t)cap);
  out->disp = take(3 * (size_t)cap);
  out->energy = take(B);
  out->forces = take(3 * (size_t)N);
  out->virial = take(9 * (size_t)B);
  out->atom_energy = take(N);
  out->f32_count = o;
  return NNHIP_OK;
}

extern "C" int nnhip_forward_dev(const nnhip_model* model, const nnhip_step_dev* st, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  TRY(check_model(model, "nnhip_forward_dev"));
  if (!st || !st->i32 || !st->f32 || !st->tail_host || !st->prepared || !st->atom_node || !st->force_node) {
    nnhip_set_error("nnhip_forward_dev: bad arguments");
    return NNHIP_E_INVALID;
  }
  const int N = st->n_atoms, B = st->n_mol, cap = st->capacity;
  nnhip_step_layout lay;
  TRY(nnhip_step_layout_of(N, B, cap, &lay));
  int32_t* I = st->i32;
  float* F = st->f32;
  int32_t *mol_ptr = I + lay.mol_ptr, *row_ptr = I + lay.row_ptr, *status = I + lay.status, *pair_ptr = I + lay.pair_ptr;
  // the node-embedding gather (edge.hip:embed_kernel) rides in the parameter check's launch: it needs the workspace addresses
  // energy_forces_impl will use (same layout function; the size / alignment checks it repeats come first here)
  EmbedJob ej = EmbedJob{nullptr, nullptr, nullptr, 0, nullptr, nullptr};
  const GraphInit no_init = GraphInit{nullptr, nullptr, 0, nullptr, 0};
  if (N >= 1 && model->n_features == NF && model->n_layers >= 1 && model->n_layers <= NNHIP_MAX_LAYERS) {
    WsInternal w;
    make_layout(N, cap, B, model->n_layers, w);
    PrepLayout pq;
    make_prep_layout(model->n_layers, pq);
    if (st->workspace && st->workspace_bytes >= w.pub.total && ((uintptr_t)st->workspace & 255) == 0 && st->prepared_bytes >= pq.total)
      ej = EmbedJob{st->z, model->node_embedding, (const float*)((const char*)st->prepared + pq.m_tab), N,
                    (float*)((char*)st->workspace + w.pub.a0), (float*)((char*)st->workspace + w.pub.m[0])};
  }
  const bool embedded = ej.a0 != nullptr;
  if (N >= 1 && N <= nnhip_graph_small_max_atoms()) {
    // a small system: the whole neighbor list in one launch (graph.hip:graph_small_kernel), then the parameter check ORs its bit
    // into the status word behind the count, then the two words leave for the host
    const int32_t* changes = nullptr;
    TRY(prepare_check_counter_impl(model, st->prepared, st->prepared_bytes, &changes, stream_, no_init, ej));
    TRY(nnhip_graph_small_dev(st->pos, st->cell, st->batch, st->z, N, B, cap, model->cutoff, mol_ptr, row_ptr, pair_ptr, st->tail_host,
                              changes, st->seq, I + lay.col, I + lay.rev, I + lay.pid, F + lay.disp, st->edge_index, model->frequencies,
                              model->n_basis, F + lay.geo, I + lay.xg, model->envelope, stream_));
    if (st->event) HIP_TRY(hipEventRecord((hipEvent_t)st->event, s));
    return energy_forces_impl(model, st->z, st->pos, st->cell, mol_ptr, row_ptr, I + lay.col, I + lay.rev, I + lay.pid, F + lay.geo,
                              I + lay.xg, F + lay.disp, N, cap, B, st->workspace, st->workspace_bytes, F + lay.energy,
                              st->want_forces ? F + lay.forces : nullptr, (st->want_forces && st->want_virial) ? F + lay.virial : nullptr,
                              F + lay.atom_energy, st->atom_node, st->force_node, st->prepared, pair_ptr + N, pair_ptr, stream_,
                            (st->flags & 1) != 0, embedded);
  }
  static const bool mol_graph_off = getenv("NNHIP_GRAPH_MOL") && atoi(getenv("NNHIP_GRAPH_MOL")) == 0;
  const int32_t* changes = nullptr;
  if ((st->flags & 1) && B >= 1 && (long)N <= (long)B * NNHIP_MOL_STAGE_MAX && !mol_graph_off) {
    // a batch of small molecules: the list by one workgroup per molecule (graph.hip:graph_mol_*_kernel), five launches instead of nine
    // (the parameter check's launch also clears status / mol_ptr / row_ptr)
    TRY(prepare_check_counter_impl(model, st->prepared, st->prepared_bytes, &changes, stream_,
                                   GraphInit{status, mol_ptr, B + 1, row_ptr, N + 1}, ej));
    TRY(nnhip_graph_mol_dev(st->pos, st->cell, st->batch, st->z, N, B, cap, model->cutoff, mol_ptr, row_ptr, pair_ptr, status,
                            I + lay.mol_scratch, 1, st->tail_host, changes, st->seq, I + lay.col, I + lay.rev, I + lay.pid,
                            F + lay.disp, F + lay.geo, I + lay.xg, stream_));
    if (st->edge_index)
      TRY(nnhip_edge_index_from_csr(row_ptr, I + lay.col, N, cap, st->edge_index, row_ptr + N, stream_));   // (hip.py passes NULL and builds it on demand)
    if (st->event) HIP_TRY(hipEventRecord((hipEvent_t)st->event, s));
    return energy_forces_impl(model, st->z, st->pos, st->cell, mol_ptr, row_ptr, I + lay.col, I + lay.rev, I + lay.pid, F + lay.geo,
                              I + lay.xg, F + lay.disp, N, cap, B, st->workspace, st->workspace_bytes, F + lay.energy,
                              st->want_forces ? F + lay.forces : nullptr, (st->want_forces && st->want_virial) ? F + lay.virial : nullptr,
                              F + lay.atom_energy, st->atom_node, st->force_node, st->prepared, pair_ptr + N, pair_ptr, stream_, true, embedded);
  }
  TRY(nnhip_graph_count_pairs_z(st->pos, st->cell, st->batch, st->z, N, B, model->cutoff, mol_ptr, row_ptr, status, pair_ptr,
                                I + lay.pair_scan, stream_));
  TRY(prepare_check_counter_impl(model, st->prepared, st->prepared_bytes, &changes, stream_, no_init, ej));
  TRY(nnhip_graph_finish_dev(st->pos, st->cell, st->batch, mol_ptr, row_ptr, pair_ptr, N, B, cap, model->cutoff, I + lay.col,
                             I + lay.rev, I + lay.pid, F + lay.disp, st->edge_index, model->frequencies, model->n_basis,
                             F + lay.geo, nullptr, nullptr, I + lay.xg, model->envelope, status, st->tail_host, changes, st->seq, stream_));
  // (the guard -- the last kernel of the neighbor list -- has written count / status / change counter / seq into the pinned slot)
  if (st->event) HIP_TRY(hipEventRecord((hipEvent_t)st->event, s));
  return energy_forces_impl(model, st->z, st->pos, st->cell, mol_ptr, row_ptr, I + lay.col, I + lay.rev, I + lay.pid, F + lay.geo,
                            I + lay.xg, F + lay.disp, N, cap, B, st->workspace, st->workspace_bytes, F + lay.energy,
                            st->want_forces ? F + lay.forces : nullptr, (st->want_forces && st->want_virial) ? F + lay.virial : nullptr,
                            F + lay.atom_energy, st->atom_node, st->force_node, st->prepared, pair_ptr + N, pair_ptr, stream_,
                            (st->flags & 1) != 0, embedded);
}

// ---- per-stage exports (include/newtonnet_hip.h, "Per-stage entry points"): thin wrappers over the launchers above ----
extern "C" size_t nnhip_filter_table_bytes(void) { return 3 * FT_PLANE * sizeof(float); }
extern "C" int nnhip_filter_tables(const float* const* edge_w, float* const* tables, int32_t n_layers, const float* freq,
                                   int32_t nb, int32_t envelope, void* stream) {
  if (!edge_w || !tables || !freq || n_layers < 1 || n_layers > NNHIP_MAX_LAYERS || nb < 1 || nb > NNHIP_MAX_NB ||
      (envelope < 0 && envelope != NNHIP_ENVELOPE_COSINE) || envelope > 64) {
    nnhip_set_error("nnhip_filter_tables: bad arguments");
    return NNHIP_E_INVALID;
  }
  return launch_filter_tables(edge_w, tables, n_layers, freq, nb, envelope, (hipStream_t)stream);
}
extern "C" int nnhip_transpose128(const float* const* src, float* const* dst, int32_t count, void* stream) {
  if (!src || !dst || count < 0 || count > 40) {
    nnhip_set_error("nnhip_transpose128: bad arguments");
    return NNHIP_E_INVALID;
  }
  return count ? launch_transposes(src, dst, count, (hipStream_t)stream) : NNHIP_OK;
}
extern "C" int nnhip_message_fwd(const float* m, const int32_t* xg, const float* table, const int32_t* row_ptr,
                                 const int32_t* col, const int32_t* pid, const float* a_in, float* msg, float* a_mid,
                                 int32_t n_atoms, void* stream) {
  if (n_atoms <= 0) return NNHIP_OK;
  return launch_msg_fwd(m, xg, table, row_ptr, col, pid, a_in, msg, a_mid, n_atoms, (hipStream_t)stream);
}
extern "C" int nnhip_message_bwd(const float* g_msg, const float* g_a, const float* m, const int32_t* xg, const float* table,
                                 const int32_t* row_ptr, const int32_t* col, const int32_t* pid, float* g_m, float* g_x,
                                 int32_t n_atoms, int32_t need_gm, void* stream) {
  if (n_atoms <= 0) return NNHIP_OK;
  return launch_msg_bwd(g_msg, g_a, m, xg, table, row_ptr, col, pid, g_m, g_x, n_atoms, need_gm != 0, (hipStream_t)stream);
}
extern "C" int nnhip_force_message_fwd(const float* phi1, const float* phi2, const float* geo, const int32_t* xg,
                                       const int32_t* row_ptr, const int32_t* col, const int32_t* pid, const float* f_in,
                                       float* f_out, int32_t n_atoms, void* stream) {
  if (n_atoms <= 0) return NNHIP_OK;
  return launch_force_fwd(f_in != nullptr, phi1, phi2, geo, row_ptr, col, pid, f_in, f_out, n_atoms, xg, (hipStream_t)stream);
}
extern "C" int nnhip_force_message_bwd(const float* gf, const float* phi1, const float* phi2, const float* geo,
                                       const int32_t* xg, const int32_t* row_ptr, const int32_t* col, const int32_t* pid,
                                       const float* f_in, float* g_h12, float* g_u, float* g_fin, int32_t n_atoms,
                                       void* stream) {
  if (n_atoms <= 0) return NNHIP_OK;
  return launch_force_bwd(f_in != nullptr, gf, phi1, phi2, geo, row_ptr, col, pid, f_in, g_h12, g_u, g_fin, n_atoms, xg,
                          (hipStream_t)stream);
}
extern "C" int nnhip_edge_embed_bwd(const float* g_x, const float* g_u, const float* geo, const float* disp, const float* pos,
                                    const float* cell, const int32_t* row_ptr, const int32_t* col, const int32_t* rev,
                                    const int32_t* mol_ptr, int32_t n_atoms, int32_t n_edges, int32_t n_mol, int32_t n_layers,
                                    float cutoff, float* g_d, float* forces, float* virial, void* stream) {
  if (n_atoms <= 0) return NNHIP_OK;
  return launch_geometry_bwd(g_x, g_u, geo, disp, pos, cell, row_ptr, col, rev, mol_ptr, n_atoms, n_edges, n_mol, n_layers,
                             cutoff, g_d, forces, virial, (hipStream_t)stream);
}
extern "C" int nnhip_node_fwd(const float* f, const float* a_mid, const float* Wu, float* q, float* a_out, const float* W0,
                              const float* b0, const float* W2, const float* b2, float* hn, float* m, int32_t n_atoms,
                              int32_t activation, void* stream) {
  NodeFwdArgs na;
  memset(&na, 0, sizeof(na));
  na.f = f; na.a_mid = a_mid; na.Wu = Wu; na.q = q; na.a_out = a_out;
  na.W0 = W0; na.b0 = b0; na.W2 = W2; na.b2 = b2; na.hn = hn; na.m = m;
  na.N = n_atoms;
  na.act = activation;
  return launch_node_fwd(na, (hipStream_t)stream);
}
extern "C" int nnhip_node_bwd(const float* g_top, const float* h_top, const float* W2T, const float* W0T, float* g_a,
                              int32_t accumulate_ga, const float* f, const float* q, const float* G_f, const float* WuT,
                              float* gf, int32_t n_atoms, int32_t activation, void* stream) {
  NodeBwdArgs nb;
  memset(&nb, 0, sizeof(nb));
  nb.g_top = g_top; nb.h_top = h_top; nb.W2T = W2T; nb.W0T = W0T; nb.g_a = g_a; nb.acc_ga = accumulate_ga;
  nb.f = f; nb.q = q; nb.G_f = G_f; nb.WuT = WuT; nb.gf = gf;
  nb.N = n_atoms;
  nb.act = activation;
  return launch_node_bwd(nb, (hipStream_t)stream);
}
extern "C" int nnhip_head_out(const float* e2, const float* w4, const float* b4, const float* scale, const float* shift,
                              const int64_t* z, const int32_t* mol_ptr, int32_t n_atoms, int32_t n_mol, int32_t activation,
                              float* atom_energy, float* g_e2, float* energy, void* stream) {
  if (n_atoms <= 0) return NNHIP_OK;
  return launch_head_out(e2, w4, b4, scale, shift, z, mol_ptr, n_atoms, n_mol, activation, atom_energy, g_e2, energy,
                         (hipStream_t)stream);
}
