/*
 * newtonnet_hip.h -- C ABI of libnewtonnet_hip.so (gfx950 / MI355X).
 *
 * The reference (THGLab/NewtonNet v2.1.0) is pure Python: the hot path has no
 * FFI of its own.  The entry points below are the operations its model forward
 * dispatches through PyTorch, cut where a maintainer would bind a native
 * extension (INTEGRATION.md shows the ctypes stub).  Each one cites the
 * reference code it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - all floating-point data is fp32, row-major, contiguous; indices are
 *     int32 inside the library and int64 where the reference API exposes them;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *     nothing synchronises the device except where stated;
 *   - the library never allocates device memory: callers own every buffer
 *     (PyTorch does, in the shipped host code);
 *   - return value 0 = success, anything else = error; nnhip_last_error()
 *     returns a static description for the calling thread.
 *   - F = n_features must be 128 (the reference's default, scripts/config.yml:30-36); nb = n_basis may be
 *     1..NNHIP_MAX_NB (default 20); other sizes return NNHIP_E_UNSUPPORTED.
 */
#ifndef NEWTONNET_HIP_H
#define NEWTONNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NNHIP_F 128
#define NNHIP_NB 20        /* the reference's default n_basis */
#define NNHIP_MAX_NB 32    /* any 1 <= n_basis <= 32 runs: only the radial-filter table builder sees the basis */
#define NNHIP_MAX_LAYERS 8
#define NNHIP_MOL_STAGE_MAX 24 /* atoms of a molecule whose node rows the molecule-resident edge kernels stage in LDS */
#define NNHIP_N_ELEMENTS 119 /* rows of node_embedding / scale / shift (z = 0..118) */
/* cutoff envelope of the edge embedding: a positive value p = PolynomialCutoff(p) (representations.py:138-171; the model
 * uses p = 9, :17; 0 means 9), NNHIP_ENVELOPE_COSINE = CosineCutoff (representations.py:177-203) */
#define NNHIP_ENVELOPE_COSINE (-1)

/* activation ids (newtonnet/layers/activations.py:5-30); 'swish' and 'silu' are both NNHIP_ACT_SILU */
enum {
  NNHIP_ACT_SILU = 0,
  NNHIP_ACT_RELU = 1,
  NNHIP_ACT_ELU = 2,
  NNHIP_ACT_LEAKY_RELU = 3,
  NNHIP_ACT_TANH = 4,
  NNHIP_ACT_SIGMOID = 5,
  NNHIP_ACT_SOFTPLUS = 6,
  NNHIP_ACT_GELU = 7,
  NNHIP_ACT_SSP = 8
};

enum {
  NNHIP_OK = 0,
  NNHIP_E_INVALID = 1,      /* bad argument */
  NNHIP_E_UNSUPPORTED = 2,  /* n_features / n_basis / n_layers outside the built kernels */
  NNHIP_E_WORKSPACE = 3,    /* workspace too small */
  NNHIP_E_HIP = 4           /* a HIP runtime call or kernel launch failed */
};

int nnhip_version(void);
/* bit 0: tooling build (compiled with extra flags, e.g. an ablation switch that changes results); the Python package refuses
 * to load such a library unless NNHIP_ALLOW_TOOLING_LIB=1 */
int nnhip_build_flags(void);
const char* nnhip_last_error(void);

/* --------------------------------------------------------------------------
 * Parameters of one model, in the reference's state_dict layout
 * (nn.Linear weights are [out][in] row-major).
 * Replaces: the nn.Module parameter storage of
 *   newtonnet/models/newtonnet.py:116-205 and newtonnet/models/output.py:88-96,
 *   newtonnet/layers/scalers.py:42-45.
 * ------------------------------------------------------------------------ */
typedef struct {
  const float* node0_w;  /* interaction_layers.l.message_nodepart.0.weight [F][F] */
  const float* node0_b;  /* ...message_nodepart.0.bias   [F] */
  const float* node2_w;  /* ...message_nodepart.2.weight [F][F] */
  const float* node2_b;  /* ...message_nodepart.2.bias   [F] */
  const float* edge_w;   /* ...message_edgepart.weight   [F][nb] */
  const float* eq1_0_w;  /* ...equiv_message1.0.weight   [F][F] */
  const float* eq1_2_w;  /* ...equiv_message1.2.weight   [F][F] */
  const float* eq2_0_w;  /* ...equiv_message2.0.weight   [F][F] */
  const float* eq2_2_w;  /* ...equiv_message2.2.weight   [F][F] */
  const float* update_w; /* ...equiv_update.weight       [F][F] */
  const float* ln_w;     /* ...layer_norm.weight         [F]  (NULL with ln_b: layer_norm=False, newtonnet.py:202-205) */
  const float* ln_b;     /* ...layer_norm.bias           [F] */
} nnhip_layer_params;

typedef struct {
  int32_t n_features; /* 128 */
  int32_t n_basis;    /* 1..NNHIP_MAX_NB (20 in the reference's configs) */
  int32_t n_layers;   /* 1..NNHIP_MAX_LAYERS */
  float cutoff;       /* Angstrom */
  const float* node_embedding; /* embedding_layers.node_embedding.weight [119][F] */
  const float* frequencies;    /* embedding_layers.edge_embedding.embedding.frequencies [nb] */
  nnhip_layer_params layer[NNHIP_MAX_LAYERS];
  const float* head0_w; /* output_layers.k.layers.0.weight [F][F] */
  const float* head0_b; /* [F] */
  const float* head2_w; /* output_layers.k.layers.2.weight [F][F] */
  const float* head2_b; /* [F] */
  const float* head4_w; /* output_layers.k.layers.4.weight [1][F] */
  const float* head4_b; /* [1] */
  const float* scale;   /* scalers.k.scale.weight [119] (NULL = 1) */
  const float* shift;   /* scalers.k.shift.weight [119] (NULL = 0) */
  int32_t activation;   /* NNHIP_ACT_* of every MLP of the model (constructor argument `activation`, newtonnet.py:30) */
  int32_t envelope;     /* cutoff envelope (see NNHIP_ENVELOPE_COSINE); 0 = the model's PolynomialCutoff(9) */
} nnhip_model;

/* --------------------------------------------------------------------------
 * Neighbor list.
 * Replaces: RadiusGraph.forward, newtonnet/layers/representations.py:57-100
 *   (all ordered pairs inside a molecule, i != j, optional single-image
 *   minimum-image shift d -= cell @ round(solve(cell^T, d)), strict ||d|| < r).
 * Contract: `batch` is non-decreasing (PyG collation order).  Edges come out
 * sorted by (molecule, i, j) -- the reference's order -- so the list is a CSR
 * over the RECEIVER i = edge_index[0].
 *
 * nnhip_graph_count: writes row_ptr[N+1] (exclusive scan of the in-degree) and
 *   mol_ptr[B+1].  status is int32[1 + ceil(n_atoms/1024)]: status[0] gets bit 1 if `batch` is not sorted (an error) and
 *   bit 8 if a molecule has more than NNHIP_MOL_STAGE_MAX atoms (information for the choice of edge kernels, not an error);
 *   the rest is scratch of the prefix scan.
 *   The caller reads E = row_ptr[N] (a device->host copy; the only sync).
 * nnhip_graph_fill: writes col[E] (sender j), rev[E] (index of the reverse
 *   edge (j,i); the edge set is symmetric), disp[E][3] = pos_i - pos_j (after
 *   the image shift) and, if non-NULL, edge_index[2][E] int64 (reference API).
 * ------------------------------------------------------------------------ */
int nnhip_graph_count(const float* pos, const float* cell, const int64_t* batch, int32_t n_atoms, int32_t n_mol,
                      float cutoff, int32_t* mol_ptr, int32_t* row_ptr, int32_t* status, void* stream);

int nnhip_graph_fill(const float* pos, const float* cell, const int64_t* batch, const int32_t* mol_ptr,
                     const int32_t* row_ptr, int32_t n_atoms, int32_t n_mol, int32_t n_edges, float cutoff,
                     int32_t* col, int32_t* rev, float* disp, int64_t* edge_index, void* stream);
/* The same neighbor list with fewer launches behind the host's edge-count round trip (the path NewtonNet.forward takes):
 * nnhip_graph_count_pairs also leaves, per row, the number of neighbors ABOVE the row's atom (the undirected pairs the row owns)
 * in pair_cnt[0 .. N); nnhip_graph_pair_scan turns them into pair_ptr[0 .. N] (exclusive scan, in place; it can run while the
 * host waits for the edge count); nnhip_graph_finish = nnhip_graph_fill + nnhip_graph_pairs' pid + nnhip_edge_embed in two
 * launches (fill; then one thread per edge: reverse edge, pair id, geometry / radial basis / table position).  Bit-identical
 * outputs to the separate entry points. */
int nnhip_graph_count_pairs(const float* pos, const float* cell, const int64_t* batch, int32_t n_atoms, int32_t n_mol,
                            float cutoff, int32_t* mol_ptr, int32_t* row_ptr, int32_t* status, int32_t* pair_cnt, void* stream);
int nnhip_graph_pair_scan(int32_t* pair_ptr, int32_t n_atoms, int32_t* scan_scratch, void* stream);
int nnhip_graph_finish(const float* pos, const float* cell, const int64_t* batch, const int32_t* mol_ptr,
                       const int32_t* row_ptr, const int32_t* pair_ptr, int32_t n_atoms, int32_t n_mol, int32_t n_edges,
                       float cutoff, int32_t* col, int32_t* rev, int32_t* pid, float* disp, int64_t* edge_index,
                       const float* frequencies, int32_t n_basis, float* geo, float* rbf, float* drbf, int32_t* xg,
                       int32_t envelope, void* stream);
/* The same two launches BEFORE the host has read the edge count: every array holds `capacity` edges (edge_index 2 x capacity
 * int64; its rows are written at the true stride, so the first 2 E entries are the contiguous [2][E] result), the kernels take the
 * count from row_ptr[n_atoms] on the device and write nothing when it exceeds `capacity` -- the caller, who learns E from its
 * read-back, then calls nnhip_graph_finish.  Fills the GPU's idle time behind the edge-count round trip. */
int nnhip_graph_finish_early(const float* pos, const float* cell, const int64_t* batch, const int32_t* mol_ptr,
                             const int32_t* row_ptr, const int32_t* pair_ptr, int32_t n_atoms, int32_t n_mol, int32_t capacity,
                             float cutoff, int32_t* col, int32_t* rev, int32_t* pid, float* disp, int64_t* edge_index,
                             const float* frequencies, int32_t n_basis, float* geo, float* rbf, float* drbf, int32_t* xg,
                             int32_t envelope, void* stream);

/* --------------------------------------------------------------------------
 * Undirected pairs.  msg = (W_e rbf) * m[i] * m[j] (newtonnet.py:211) is symmetric under i <-> j, and so is everything
 * equiv_message1/2 compute from it (newtonnet.py:218,222): nnhip_energy_forces evaluates msg / hidden / phi once per
 * undirected pair.  pid[e] = pair row of directed edge e (pairs numbered in CSR order of their i < j edge);
 * pair_ptr[N+1] = first pair owned by each row.  n_edges must be even (symmetric edge set).
 * ------------------------------------------------------------------------ */
int nnhip_graph_pairs(const int32_t* row_ptr, const int32_t* col, const int32_t* rev, int32_t n_atoms, int32_t n_edges,
                      int32_t* pair_ptr, int32_t* pid, int32_t* scan_scratch /* int32[ceil(n_atoms/1024)] */, void* stream);

/* --------------------------------------------------------------------------
 * O(N) variant of the neighbor list for ONE large orthorhombic periodic box (BASELINE config 5; the reference's
 * all-pairs build, representations.py:74-85, needs O(N^2) memory and cannot run it).  Bit-identical output to
 * nnhip_graph_count/fill (same predicate, same displacement arithmetic, same edge order); candidates are pruned
 * with a grid of cells >= cutoff wide.  box_len_host = the three diagonal cell entries (HOST pointer); every
 * length must be >= 3 x cutoff (else NNHIP_E_UNSUPPORTED: use the all-pairs entry points).
 * scratch: nnhip_graph_cells_scratch_bytes() bytes, kept between the count and fill calls.
 * ------------------------------------------------------------------------ */
size_t nnhip_graph_cells_scratch_bytes(int32_t n_atoms, const float* box_len_host, float cutoff);
int nnhip_graph_count_cells(const float* pos, const float* cell, int32_t n_atoms, float cutoff,
                            const float* box_len_host, void* scratch, int32_t* mol_ptr, int32_t* row_ptr, void* stream);
int nnhip_graph_fill_cells(const float* pos, const float* cell, int32_t n_atoms, int32_t n_edges, float cutoff,
                           const float* box_len_host, void* scratch, const int32_t* row_ptr, int32_t* col, int32_t* rev,
                           float* disp, int64_t* edge_index, void* stream);
/* The cell-list counterparts of nnhip_graph_count_pairs / nnhip_graph_finish (same outputs as the separate entry points) */
int nnhip_graph_count_cells_pairs(const float* pos, const float* cell, int32_t n_atoms, float cutoff, const float* box_len_host,
                                  void* scratch, int32_t* mol_ptr, int32_t* row_ptr, int32_t* pair_cnt, void* stream);
int nnhip_graph_finish_cells(const float* pos, const float* cell, int32_t n_atoms, int32_t n_edges, float cutoff,
                             const float* box_len_host, void* scratch, const int32_t* row_ptr, const int32_t* pair_ptr,
                             int32_t* col, int32_t* rev, int32_t* pid, float* disp, int64_t* edge_index,
                             const float* frequencies, int32_t n_basis, float* geo, float* rbf, float* drbf, int32_t* xg,
                             int32_t envelope, void* stream);

/* --------------------------------------------------------------------------
 * Verlet-skin reuse of a neighbor list in an MD loop (caller: MLAseCalculator.calculate,
 * newtonnet/utils/ase_interface.py:52-81, which rebuilds the O(N^2) graph every step).
 * nnhip_edge_disp: disp[e] = pos_i - pos_j (+ the reference's single-image shift) for the edges of a list built
 *   earlier with cutoff + skin (edge_index[2][E] int64 from nnhip_graph_fill).  nnhip_edge_embed with the real cutoff
 *   then points candidates with r >= cutoff at all-zero filter rows, so they contribute exactly nothing.
 * ------------------------------------------------------------------------ */
int nnhip_edge_disp(const float* pos, const float* cell, const int64_t* batch, const int64_t* edge_index, int32_t n_edges,
                    float* disp, void* stream);
/* nnhip_edge_disp followed by nnhip_edge_embed, in one launch (the per-step refresh of a reused candidate list) */
int nnhip_edge_refresh(const float* pos, const float* cell, const int64_t* batch, const int64_t* edge_index, int32_t n_edges,
                       float cutoff, const float* frequencies, int32_t n_basis, float* disp, float* geo, float* rbf, float* drbf,
                       int32_t* xg, int32_t envelope, void* stream);

/* --------------------------------------------------------------------------
 * Species check.  The reference indexes nn.Embedding(119, F) / the scale and shift tables with z and raises
 * IndexError for z outside 0..118 (newtonnet.py:142, scalers.py:55-58); the kernels would read out of bounds.
 * ORs 2 into status[0] (the word nnhip_graph_count ORs 1 into for a bad batch vector; the host reads it back
 * together with the edge count) when any z[i] is outside [0, NNHIP_N_ELEMENTS).
 * ------------------------------------------------------------------------ */
int nnhip_check_species(const int64_t* z, int32_t n_atoms, int32_t* status, void* stream);

/* --------------------------------------------------------------------------
 * Edge embedding.
 * Replaces: ScaledNorm.forward (representations.py:118-133), PolynomialCutoff
 *   p=9 (:155-171), RadialBesselLayer.forward (:223-235) and their product (:41).
 * geo[E][4]  = (ux, uy, uz, r)    dir_edge and |disp|
 * rbf[E][nb] = env(x) * sin(w_n x)/x,  x = r/cutoff        (= dist_edge; may be NULL)
 * drbf[E][nb]= d rbf / d x                                  (may be NULL)
 * xg[E][2]   = (g0, bits of u): x = (g0 + u) / FT_G, the position of the edge on the radial-filter table that
 *              nnhip_energy_forces interpolates instead of contracting rbf with message_edgepart.weight per edge
 *              (may be NULL).  A candidate that fails the neighbor predicate `|disp| < cutoff` (fp32, as
 *              representations.py:96) gets the all-zero filter row and is masked out of the force kernels.
 * ------------------------------------------------------------------------ */
int nnhip_edge_embed(const float* disp, int32_t n_edges, float cutoff, const float* frequencies, int32_t n_basis,
                     float* geo, float* rbf, float* drbf, int32_t* xg, int32_t envelope, void* stream);

/* --------------------------------------------------------------------------
 * Whole hot path: energy and forces (= -dE/dpos) for a batch.
 * Replaces: NewtonNet.forward after the graph build, i.e.
 *   EmbeddingNet.forward newtonnet/models/newtonnet.py:139-161 (node part),
 *   InteractionNet.forward :207-231 for every layer,
 *   EnergyOutput.forward output.py:98-100, ScaleShift.forward scalers.py:47-59,
 *   EnergyAggregator.forward output.py:245-247,
 *   and the reverse sweep torch.autograd.grad performs for
 *   GradientForceOutput (output.py:66-73,109-113), written out analytically.
 *
 * workspace: nnhip_workspace_bytes(...) bytes, 256-byte aligned.  Per-layer
 * intermediates stay in it after the call; nnhip_workspace_layout() reports
 * where (used by the parity tests).
 * Optional outputs (may be NULL): atom_energy[N], atom_node[N][F],
 * force_node[N][3][F], virial[B][3][3] (= -dE/d strain, output.py:154-165).
 * forces may be NULL (energy only: forward sweep only).  pos / cell are read only for the virial (may be NULL
 * otherwise); with periodic cells the virial follows the reference's own strain formula, including the
 * `cell @ n` image shift of representations.py:93.
 * ------------------------------------------------------------------------ */
size_t nnhip_workspace_bytes(int32_t n_atoms, int32_t n_edges, int32_t n_mol, int32_t n_layers);

typedef struct {
  /* byte offsets into the workspace; per-layer arrays indexed by layer */
  size_t m[NNHIP_MAX_LAYERS];      /* [N][F]   message_nodepart output */
  size_t hn[NNHIP_MAX_LAYERS];     /* [N][F]   message_nodepart hidden pre-activation */
  size_t msg[NNHIP_MAX_LAYERS];    /* [P][F]   message, one row per undirected pair (P = E/2, row pid[e]) */
  size_t h12[NNHIP_MAX_LAYERS];    /* 2 x [pad32(P)][F] scratch between the equiv_message{1,2} forward and its adjoint, private to
                                      the MLP kernels: the hidden pre-activations h, row-major, up to 26 624 pair rows; above that
                                      silu'(h) in MFMA-fragment order (all the adjoint needs of h) */
  size_t phi1[NNHIP_MAX_LAYERS];   /* [P][F] */
  size_t phi2[NNHIP_MAX_LAYERS];   /* [P][F] */
  size_t a_mid[NNHIP_MAX_LAYERS];  /* [N][F]   atom_node after the invariant update */
  /* hn[0] is not written: layer 0's message_nodepart is evaluated per element (its adjoint is never needed) */
  size_t a_out[NNHIP_MAX_LAYERS];  /* [N][F]   atom_node after the layer (last layer: only when atom_node == NULL) */
  size_t f_out[NNHIP_MAX_LAYERS];  /* [N][3][F] force_node after the layer (last layer: only when force_node == NULL) */
  size_t q[NNHIP_MAX_LAYERS];      /* [N][3][F] equiv_update(force_node) */
  size_t a0;                       /* [N][F]   embedded atom_node */
  size_t e1, e2;                   /* [N][F]   head hidden pre-activations */
  size_t g_x;                      /* [L][E]   dE/dx per layer */
  size_t g_u;                      /* [L][E][3] dE/d dir per layer (only [E][4] rows: gx,guy..) */
  size_t g_a;                      /* [N][F]   running dE/d atom_node */
  size_t g_f;                      /* [N][3][F] running dE/d force_node */
  size_t total;
} nnhip_ws_layout;

int nnhip_workspace_layout(int32_t n_atoms, int32_t n_edges, int32_t n_mol, int32_t n_layers, nnhip_ws_layout* out);

int nnhip_energy_forces(const nnhip_model* model, const int64_t* z, const float* pos, const float* cell,
                        const int32_t* mol_ptr,
                        const int32_t* row_ptr, const int32_t* col, const int32_t* rev, const int32_t* pid, const float* geo,
                        const int32_t* xg, const float* disp, int32_t n_atoms, int32_t n_edges,
                        int32_t n_mol, void* workspace, size_t workspace_bytes, float* energy, float* forces,
                        float* virial, float* atom_energy, float* atom_node, float* force_node,
                        const void* prepared, void* stream);

/* The same step queued BEFORE the host knows the edge count: no device->host round trip inside a step, so a slow host costs
 * nothing (the reference has no counterpart: RadiusGraph.forward, representations.py:57-100, is synchronous by construction;
 * the caller it serves is still NewtonNet.forward, newtonnet.py:74-104).
 *   nnhip_graph_finish_dev   = nnhip_graph_finish_early into arrays of `capacity` edges (even, > 0) + a guard: when the count
 *       (row_ptr[n_atoms]) exceeds the capacity nothing was filled and the guard EMPTIES the graph on the device
 *       (every row becomes [count, count), pair_ptr all zero; graph.hip:graph_guard_kernel), so that the step below runs on zero edges, inside the arrays; likewise when the status word
 *       of nnhip_graph_count_pairs / nnhip_check_species carries bit 1 or 2 (the synchronous path raises on those BEFORE it
 *       runs the step: a broken batch vector can give an edge set without reverse edges).
 *   nnhip_energy_forces_dev  = nnhip_energy_forces with n_edges := capacity (array / workspace sizes,
 *       nnhip_workspace_bytes(n_atoms, capacity, ...)) and the true number of undirected pairs read on the device from
 *       *n_pairs_dev (= &pair_ptr[n_atoms]).
 * The host copies (row_ptr[n_atoms], status) out asynchronously BEFORE nnhip_graph_finish_dev and looks at them whenever it
 * likes (NewtonNet.forward: when a result of the call is first touched, or when the next call starts): count > capacity or a
 * stale prepared block => repeat the step the ordinary way; status bits 1 / 2 => the reference's ValueError / IndexError.
 * Species outside [0, 118] never index a table (the kernels clamp them; the status bit reports them). */
/* nnhip_graph_count_pairs with the species check of nnhip_check_species riding in its molecule-extent kernel (z may be NULL) and
 * pair_ptr scanned in the same two launches as row_ptr (pair_scan_scratch: n_atoms / 1024 + 1 ints) -- what the deferred step
 * uses instead of nnhip_graph_count_pairs + nnhip_check_species + nnhip_graph_pair_scan (three launches less). */
int nnhip_graph_count_pairs_z(const float* pos, const float* cell, const int64_t* batch, const int64_t* z, int32_t n_atoms,
                              int32_t n_mol, float cutoff, int32_t* mol_ptr, int32_t* row_ptr, int32_t* status,
                              int32_t* pair_ptr, int32_t* pair_scan_scratch, void* stream);
int nnhip_graph_finish_dev(const float* pos, const float* cell, const int64_t* batch, const int32_t* mol_ptr,
                           int32_t* row_ptr, int32_t* pair_ptr, int32_t n_atoms, int32_t n_mol, int32_t capacity,
                           float cutoff, int32_t* col, int32_t* rev, int32_t* pid, float* disp, int64_t* edge_index,
                           const float* frequencies, int32_t n_basis, float* geo, float* rbf, float* drbf, int32_t* xg,
                           int32_t envelope, const int32_t* status, int32_t* tail_host /* 4 ints, pinned host memory, or NULL */,
                           const int32_t* changes /* nnhip_prepare_check_counter's counter, or NULL */,
                           int32_t seq /* stored into tail_host[3] after the three words */, void* stream);
int nnhip_energy_forces_dev(const nnhip_model* model, const int64_t* z, const float* pos, const float* cell,
                            const int32_t* mol_ptr,
                            const int32_t* row_ptr, const int32_t* col, const int32_t* rev, const int32_t* pid, const float* geo,
                            const int32_t* xg, const float* disp, int32_t n_atoms, int32_t capacity,
                            int32_t n_mol, void* workspace, size_t workspace_bytes, float* energy, float* forces,
                            float* virial, float* atom_energy, float* atom_node, float* force_node,
                            const void* prepared, const int32_t* n_pairs_dev, void* stream);

/* The neighbor list of a small system (1 .. 1024 atoms; nnhip_forward_dev uses it up to nnhip_graph_small_max_atoms() = 128, where one
 * workgroup still beats fourteen parallel launches) in ONE launch: everything nnhip_graph_count_pairs,
 * nnhip_check_species, nnhip_graph_pair_scan and nnhip_graph_finish_dev do, as the phases of one workgroup (same list, bit for
 * bit).  tail[0] = the true edge count, tail[1] = the status bits, tail[2] = *changes; when the count exceeds `capacity` or a status bit 1 / 2 is
 * set, row_ptr / pair_ptr come back all-zero (an emptied graph, as nnhip_graph_finish_dev leaves one).  Used by nnhip_forward_dev. */
int nnhip_graph_small_dev(const float* pos, const float* cell, const int64_t* batch, const int64_t* z, int32_t n_atoms,
                          int32_t n_mol, int32_t capacity, float cutoff, int32_t* mol_ptr, int32_t* row_ptr,
                          int32_t* pair_ptr, int32_t* tail /* 4 ints; may be pinned host memory */,
                          const int32_t* changes /* or NULL */, int32_t seq /* -> tail[3], last */, int32_t* col, int32_t* rev, int32_t* pid, float* disp,
                          int64_t* edge_index, const float* frequencies, int32_t n_basis, float* geo, int32_t* xg,
                          int32_t envelope, void* stream);
int nnhip_graph_small_max_atoms(void);

/* The neighbor list of a batch of small molecules, one workgroup per molecule (graph.hip:graph_mol_count_kernel,
 * graph_mol_fill_kernel): what nnhip_graph_count_pairs_z + nnhip_graph_finish_dev do in eight launches, in five (init, molecule
 * extents + species check, counts, fill, the guard; four with initialised != 0: the caller has zeroed status[0] and
 * mol_ptr[0 .. n_mol]).  Serves molecules of up to 1024 atoms; nnhip_forward_dev uses it when flags bit 0 says the batch holds small
 * molecules only.  status: one int32 (bits 1 / 2 as nnhip_graph_count / nnhip_check_species; 8: a molecule above
 * NNHIP_MOL_STAGE_MAX atoms; 16: a molecule above 1024 atoms -- the graph is emptied).  scratch: 2 n_mol + n_mol / 1024 + 4 ints
 * (any content).  tail_host / changes / seq, the emptied graph and the list itself: as nnhip_graph_finish_dev (bit for bit the
 * same list; geo / xg as nnhip_edge_embed without rbf). */
int nnhip_graph_mol_dev(const float* pos, const float* cell, const int64_t* batch, const int64_t* z, int32_t n_atoms,
                        int32_t n_mol, int32_t capacity, float cutoff, int32_t* mol_ptr, int32_t* row_ptr, int32_t* pair_ptr,
                        int32_t* status, int32_t* scratch, int32_t initialised, int32_t* tail_host, const int32_t* changes,
                        int32_t seq, int32_t* col, int32_t* rev, int32_t* pid, float* disp, float* geo, int32_t* xg, void* stream);
/* edge_index [2][n_edges] int64 (RadiusGraph's API array, representations.py:98-100) from the CSR list: row 0 = the receiver of
 * each edge, row 1 = col.  n_edges_dev (optional): the count is read on the device, n_edges is then the capacity of the grid. */
int nnhip_edge_index_from_csr(const int32_t* row_ptr, const int32_t* col, int32_t n_atoms, int32_t n_edges,
                              int64_t* edge_index, const int32_t* n_edges_dev, void* stream);

/* The whole deferred step in ONE call (what NewtonNet.forward issues in its steady state: the ~8 host calls of the pieces above
 * cost a small molecule more than its kernels): nnhip_prepare_check_counter, the neighbor list (nnhip_graph_count_pairs_z +
 * nnhip_graph_finish_dev, or nnhip_graph_small_dev for small systems) whose last kernel stores (edge count, status, change counter,
 * seq) into tail_host[4] (pinned host memory; the caller polls tail_host[3] for its seq, or waits on the optional `event`), and
 * nnhip_energy_forces_dev.  Everything per-step lives in caller-allocated arenas laid out by nnhip_step_layout. */
typedef struct {
  size_t i32_count, f32_count;   /* elements of the two arenas (int32 / float32, both 256-byte aligned by the caller) */
  /* int32 arena */
  size_t mol_ptr, row_ptr, status, pair_ptr, pair_scan, tail /* 2 ints */, mol_scratch /* 2 n_mol + n_mol / 1024 + 4 ints */, xg, col, rev, pid;
  /* float32 arena: edge geometry, then the small outputs */
  size_t geo, disp, energy, forces, virial, atom_energy;
} nnhip_step_layout;
int nnhip_step_layout_of(int32_t n_atoms, int32_t n_mol, int32_t capacity, nnhip_step_layout* out);
typedef struct {
  const int64_t* z;        /* [N] */
  const float* pos;        /* [N][3] */
  const float* cell;       /* [B][3][3] */
  const int64_t* batch;    /* [N] */
  int32_t n_atoms, n_mol, capacity /* even, > 0 */, want_forces, want_virial;
  int32_t seq;             /* the caller's sequence number of this step: stored into tail_host[3] AFTER the three words */
  int32_t flags, pad_;     /* bit 0: every molecule of the batch is expected to have at most NNHIP_MOL_STAGE_MAX atoms (the status
                              word of the previous batch of this shape had bit 8 clear): the molecule-resident edge kernels may run */
  int32_t* i32;            /* arena of nnhip_step_layout.i32_count ints */
  float* f32;              /* arena of nnhip_step_layout.f32_count floats */
  int64_t* edge_index;     /* [2 * capacity] or NULL (what newtonnet_amd/hip.py passes: it calls nnhip_edge_index_from_csr when a
                              caller asks for the array); rows at stride = the TRUE edge count */
  float* atom_node;        /* [N][F] */
  float* force_node;       /* [N][3][F] */
  void* workspace;         /* nnhip_workspace_bytes(N, capacity, B, L) */
  size_t workspace_bytes;
  void* prepared;          /* nnhip_prepared_bytes(L), filled by nnhip_prepare */
  size_t prepared_bytes;
  int32_t* tail_host;      /* pinned host memory, 4 ints: (edge count, status, change counter of the prepared block: non-zero =
                              a parameter changed since nnhip_prepare, seq) -- written by a kernel's own stores (zero-copy), seq
                              last with release semantics: a host that reads tail_host[3] == seq may read the other three */
  void* event;             /* optional hipEvent_t recorded behind that kernel (NULL: none -- a record is a marker packet in the
                              stream, ~6 us of bubble per step; NewtonNet.forward polls tail_host[3] instead) */
} nnhip_step_dev;
int nnhip_forward_dev(const nnhip_model* model, const nnhip_step_dev* step, void* stream);

/* nnhip_energy_forces for a caller that kept pair_ptr[n_atoms + 1] (the scan of the per-row pair counts): the row kernels then
 * find the split of a row into "pairs the other endpoint owns | pairs this row owns" with two scalar loads instead of a ballot
 * over its cols.  Same results.  flags bit 0: no molecule of the batch has more than NNHIP_MOL_STAGE_MAX atoms (bit 8 of the
 * status word nnhip_graph_count* left is clear): large batches then run force_fwd one workgroup per molecule with the molecule's
 * node rows staged in LDS (edge.hip:force_fwd_mol_kernel; same sums in a different order than the row form, ~1e-7 relative). */
int nnhip_energy_forces_pp(const nnhip_model* model, const int64_t* z, const float* pos, const float* cell,
                           const int32_t* mol_ptr,
                           const int32_t* row_ptr, const int32_t* col, const int32_t* rev, const int32_t* pid, const float* geo,
                           const int32_t* xg, const float* disp, int32_t n_atoms, int32_t n_edges,
                           int32_t n_mol, void* workspace, size_t workspace_bytes, float* energy, float* forces,
                           float* virial, float* atom_energy, float* atom_node, float* force_node,
                           const void* prepared, const int32_t* pair_ptr, int32_t flags, void* stream);

/* Parameter-only preparation (transposed weights for the reverse sweep, radial-filter tables, layer 0's
 * message_nodepart per element): what the reference gets for free from nn.Module state.  `prepared` is a caller-owned
 * 256-byte-aligned block of nnhip_prepared_bytes(n_layers); fill it with nnhip_prepare whenever the parameters may have
 * changed (the Python mirror does it on every forward, on the stream, while the host waits for the edge count) and pass it
 * to nnhip_energy_forces.  prepared == NULL: nnhip_energy_forces rebuilds the block inside its workspace on every call. */
size_t nnhip_prepared_bytes(int32_t n_layers);
int nnhip_prepare(const nnhip_model* model, void* prepared, size_t prepared_bytes, void* stream);
/* Has any parameter changed since `prepared` was last FILLED?  One launch: every parameter tensor of `model` is compared bit for
 * bit with the snapshot nnhip_prepare took when it filled the block, and `bit` is OR-ed into *status (device int32) when
 * something differs.  Compare only: the snapshot moves in nnhip_prepare alone, so a change stays visible until the block has
 * really been refilled (a caller that fails between the check and the refill cannot lose it).  A caller that keeps one block
 * per model runs this every call and calls nnhip_prepare only when the bit comes back set.  Replaces nothing in the
 * reference: torch modules keep no derived state. */
int nnhip_prepare_check(const nnhip_model* model, void* prepared, size_t prepared_bytes, int32_t* status, int32_t bit,
                        void* stream);
/* The same comparison reporting through an int32 counter INSIDE the block: zero after nnhip_prepare, one or more up for every
 * check that finds a difference (so it stays non-zero until the block is refilled).  Nothing of the caller's has to be
 * initialised for it.  *counter (optional) receives the counter's device address: the deferred step hands the value to the host
 * next to the edge count (nnhip_graph_finish_dev / nnhip_graph_small_dev, `changes`). */
int nnhip_prepare_check_counter(const nnhip_model* model, void* prepared, size_t prepared_bytes, const int32_t** counter,
                                void* stream);

/* --------------------------------------------------------------------------
 * One dense 128 -> 128 linear on the matrix cores (fp32 MFMA, exact fp32):
 *   C[M][128] = epilogue( prologue(A)[M][128] . W^T ),   W = [128 out][128 in] (nn.Linear layout)
 * Replaces: torch.nn.functional.linear for the F x F layers of the path
 *   (message_nodepart newtonnet.py:181-185, equiv_message1/2 :188-197, equiv_update :199,
 *   EnergyOutput.layers output.py:90-95) and the x W products of their adjoints (pass W^T).
 * prologue: 0 none, 1 A <- silu(A).   epilogue: 0 store, 1 + bias[n], 2 * silu'(H[m][n]), 3 C += result.
 * lda/ldc/ldh are row strides in floats (>= 128).  C may alias A only exactly (same pointer, ldc == lda).
 * ------------------------------------------------------------------------ */
int nnhip_linear128(const float* A, int32_t lda, const float* W, float* C, int32_t ldc, const float* bias,
                    const float* H, int32_t ldh, int32_t M, int32_t prologue, int32_t epilogue, void* stream);

/* --------------------------------------------------------------------------
 * Fused two-layer edge MLP (Linear -> SiLU -> Linear, no bias) and its adjoint; the hidden tile stays in registers.
 * Replaces: equiv_message1 / equiv_message2 (newtonnet.py:188-197, called at :218,:222) -- ~87 % of the FLOPs.
 *   mode 0 (forward):  H = X W1^T (written: the adjoint needs it),  Y = silu(H) W2^T
 *   mode 1 (adjoint):  G = (X W1^T) * silu'(H) (H read),            Y = G W2^T   (accumulate != 0: Y += ...)
 *                      call with X = g_phi, W1 = V2^T, W2 = V1^T to get Y = g_msg.
 * W1, W2: [128][128] row-major.  ldx / ldh / ldy: row strides in floats (>= 128, multiples of 4).
 * ------------------------------------------------------------------------ */
int nnhip_mlp128(const float* X, int32_t ldx, const float* W1, const float* W2, float* H, int32_t ldh, float* Y,
                 int32_t ldy, int32_t M, int32_t mode, int32_t accumulate, void* stream);

/* --------------------------------------------------------------------------
 * direct_force head.
 * Replaces: DirectForceOutput.forward (newtonnet/models/output.py:115-132) + ScaleShift (scalers.py:55-56):
 *   d = Linear(silu(Linear(silu(Linear(atom_node)))));  out[i][k] = scale[z_i] * < d[i], force_node[i][k] >
 * w0/w2/w4: output_layers.k.layers.{0,2,4}.weight [F][F], b0/b2/b4 biases [F]; scale: scalers.k.scale.weight [119]
 * (NULL = 1).  atom_node [N][F] and force_node [N][3][F] are the outputs of nnhip_energy_forces.
 * scratch: 3 * n_atoms * F floats.
 * ------------------------------------------------------------------------ */
int nnhip_direct_force(const float* atom_node, const float* force_node, const int64_t* z, const float* w0, const float* b0,
                       const float* w2, const float* b2, const float* w4, const float* b4, const float* scale,
                       int32_t activation, int32_t n_atoms, float* scratch, float* out, void* stream);

/* --------------------------------------------------------------------------
 * Differentiable building blocks of the train-mode forward (both are linear maps and each other's adjoints, so
 * torch.autograd can differentiate through them twice: force-loss training, output.py:66-73 + trainer.py:307-309).
 * Replaces: torch_geometric.utils.scatter(..., reduce='sum') (newtonnet.py:214,226; output.py:246) and the
 *   index gathers m[edge_index[k]], force_node[edge_index[1]] (newtonnet.py:211,223).
 *   segment_sum: out[i][:] = sum_{e in [row_ptr[i], row_ptr[i+1])} x[e][:]   (CSR rows; deterministic, no atomics)
 *   gather_rows: out[e][:] = x[idx[e]][:]
 * width = floats per row (even).
 * ------------------------------------------------------------------------ */
int nnhip_segment_sum(const float* x, const int32_t* row_ptr, int32_t n_rows, int32_t width, float* out, void* stream);
int nnhip_gather_rows(const float* x, const int32_t* idx, int32_t n_out, int32_t width, float* out, void* stream);

/* ==========================================================================
 * Per-stage entry points (last row) and TRAINING.
 *
 * The stages nnhip_energy_forces runs internally, exported one by one, plus the kernels that differentiate them
 * once more.  Training (newtonnet/train/trainer.py:299-313: loss.backward() through the autograd force of
 * newtonnet/models/output.py:66-73, create_graph=True) is evaluated as "tangent over reverse": with c_b = dL/dE_b and
 * d = dL/dF,   dL/dtheta = sum_b c_b dE_b/dtheta - D_d[grad_theta E_tot],   i.e. the reverse sweep is differentiated in
 * forward (tangent) mode along v = -d with the reverse seed carried as the dual number 1 + eps c_b.  Every stage below has a
 * value form (*_fwd / *_bwd = its adjoint) and a tangent form (*_tan_fwd / *_tan_bwd); all are deterministic (segmented
 * sums over the receiver CSR, pair rows written by the lower endpoint, no float atomics).  The host side that strings
 * them together is newtonnet_amd/train_fused.py; tests/tangent_ref.py states the same sweeps in fp64.
 * Shapes: N atoms, E directed edges, P = E/2 pairs (row pid[e]); node rows [N][F], [N][3][F]; pair rows [P][F].
 * ========================================================================== */

/* atom_node = Embedding[z]  (newtonnet.py:142) */
int nnhip_embed(const int64_t* z, const float* table, int32_t n_atoms, float* out, void* stream);

/* Radial-filter tables of `n_layers` layers (message_edgepart applied to the Bessel basis, newtonnet.py:186,210): for each
 * layer three planes of FT_ROWS rows of F floats over the nodes x_g = g / FT_G (FT_G = 3072 intervals of x = r/cutoff; row =
 * g + 1) -- value, secant slope to the next node (differenced in fp64), d/dx; tables[l] needs nnhip_filter_table_bytes() bytes.
 * The message kernels interpolate (value: 4-point Lagrange on the value plane; value and derivative: cubic Hermite from four
 * rows) instead of contracting rbf per edge. */
size_t nnhip_filter_table_bytes(void);
int nnhip_filter_tables(const float* const* edge_w_host_array, float* const* tables_host_array, int32_t n_layers,
                        const float* frequencies, int32_t n_basis, int32_t envelope, void* stream);

/* out[k] = in[k]^T for `count` <= 40 [128][128] matrices (weights for the adjoint products) */
int nnhip_transpose128(const float* const* src_host_array, float* const* dst_host_array, int32_t count, void* stream);

/* K1: message + invariant aggregation (newtonnet.py:210-215) and its adjoint.
 *   fwd: msg[p] = eps(x_e) m_i m_j;  a_mid[i] = a_in[i] + sum_{e in row i} msg
 *   bwd: G = g_msg[p] + g_a[i] + g_a[j];  g_m[i] = sum_e G eps m_j;  g_x[e] = <G m_i m_j, d eps/dx> (pair owner's edge) */
int nnhip_message_fwd(const float* m, const int32_t* xg, const float* table, const int32_t* row_ptr, const int32_t* col,
                      const int32_t* pid, const float* a_in, float* msg, float* a_mid, int32_t n_atoms, void* stream);
int nnhip_message_bwd(const float* g_msg, const float* g_a, const float* m, const int32_t* xg, const float* table,
                      const int32_t* row_ptr, const int32_t* col, const int32_t* pid, float* g_m, float* g_x,
                      int32_t n_atoms, int32_t need_gm, void* stream);

/* K2: equivariant messages + aggregation (newtonnet.py:219-227) and its adjoint.  f_in NULL = first layer (force_node == 0).
 *   fwd: f_out[i][k] = f_in[i][k] + sum_e phi1[p] u_e[k] + phi2[p] f_in[j][k]
 *   bwd: g_h12[p] = (g_phi1 | g_phi2) (both directions summed), g_u[e][k] = <gf_i[k], phi1[p]>,
 *        g_fin[i][k] = gf[i][k] + sum_e phi2[p] gf[j][k] */
int nnhip_force_message_fwd(const float* phi1, const float* phi2, const float* geo, const int32_t* xg,
                            const int32_t* row_ptr, const int32_t* col, const int32_t* pid, const float* f_in,
                            float* f_out, int32_t n_atoms, void* stream);
int nnhip_force_message_bwd(const float* gf, const float* phi1, const float* phi2, const float* geo, const int32_t* xg,
                            const int32_t* row_ptr, const int32_t* col, const int32_t* pid, const float* f_in,
                            float* g_h12, float* g_u, float* g_fin, int32_t n_atoms, void* stream);

/* Adjoint of the edge embedding: g_x[L][E], g_u[L][E][4] of all layers -> g_d[E][4] -> forces[N][3] (= -dE/dpos) and,
 * optionally, virial[B][3][3] (output.py:154-165; needs disp / pos / cell / mol_ptr). */
int nnhip_edge_embed_bwd(const float* g_x, const float* g_u, const float* geo, const float* disp, const float* pos,
                         const float* cell, const int32_t* row_ptr, const int32_t* col, const int32_t* rev,
                         const int32_t* mol_ptr, int32_t n_atoms, int32_t n_edges, int32_t n_mol, int32_t n_layers,
                         float cutoff, float* g_d, float* forces, float* virial, void* stream);

/* Row-local node stages between two edge phases (newtonnet.py:229-231 and, for the next layer, :181-185,209).
 *   node_fwd: q_k = f_k Wu^T; a_out = a_mid + sum_k f_k q_k; [hn = a_out W0^T + b0; m = act(hn) W2^T + b2 when W0 != NULL]
 *   node_bwd: [g_hn = (g_top W2T^T) act'(h_top); g_a (+)= g_hn W0T^T when W2T != NULL];
 *             [gf_k = G_f,k + g_a q_k + (g_a f_k) WuT^T when WuT != NULL]     (W*T = transposed weights) */
int nnhip_node_fwd(const float* f, const float* a_mid, const float* Wu, float* q, float* a_out, const float* W0,
                   const float* b0, const float* W2, const float* b2, float* hn, float* m, int32_t n_atoms,
                   int32_t activation, void* stream);
int nnhip_node_bwd(const float* g_top, const float* h_top, const float* W2T, const float* W0T, float* g_a,
                   int32_t accumulate_ga, const float* f, const float* q, const float* G_f, const float* WuT, float* gf,
                   int32_t n_atoms, int32_t activation, void* stream);

/* Energy head tail (output.py:98-100 last linear, scalers.py:55-58, output.py:246) and the reverse seed:
 *   atom_energy[i] = (<act(e2_i), w4> + b4) scale[z_i] + shift[z_i];  energy[b] = sum;  g_e2[i] = scale w4 act'(e2)  (may be NULL) */
int nnhip_head_out(const float* e2, const float* w4, const float* b4, const float* scale, const float* shift,
                   const int64_t* z, const int32_t* mol_ptr, int32_t n_atoms, int32_t n_mol, int32_t activation,
                   float* atom_energy, float* g_e2, float* energy, void* stream);

/* Fused Linear -> act -> Linear with every option (nnhip_mlp128 is the plain subset):
 *   mode 0 FWD : H = X W1^T + b1 (stored);  Y = act(H) W2^T + b2
 *   mode 1 BWD : T = X W1^T;  Y (+)= (T act'(H)) W2^T
 *   mode 2 TAN : as BWD, and T is stored            -- tangent of FWD (X = dX: T = dH, Y = dY), or BWD keeping its hidden product
 *   mode 3 TAN2: dT = X W1^T;  G = dT act'(H) + T2 act''(H) Hd (stored);  Y (+)= G W2^T      -- tangent of BWD
 * H / T / T2 / Hd / G share the row pitch ldh. */
typedef struct {
  const float* X; int32_t ldx;
  const float* W1; const float* W2; const float* b1; const float* b2;
  float* H; int32_t ldh;
  float* Y; int32_t ldy;
  int32_t M, mode, accumulate, activation;
  float* T; const float* T2; const float* Hd; float* G;
  /* optional: split-f16 images of W1 / W2 (nnhip_weight_images; both or neither).  With them the row-local form of the kernel
   * (M <= 49 152 rows) takes the split-f16 product form instead of v_mfma_f32_32x32x2_f32; results agree to fp32 rounding. */
  const void* W1_image; const void* W2_image;
  /* 0: fp32-grade products (the default).  1: the bf16 compute mode of training under torch.autocast(bfloat16) (BASELINE configs[2];
   * the reference has no bf16, newtonnet/layers/precision.py:3-13): both operands of every product rounded to bf16, ONE
   * v_mfma_f32_32x32x16_bf16 per 16 k-values instead of three f16 ones, fp32 accumulation, fp32 inputs / outputs; bias-free SiLU
   * MLPs whose images were written by nnhip_weight_images_bf16. */
  int32_t precision; int32_t pad_;
} nnhip_mlp_desc;
int nnhip_mlp128_ex(const nnhip_mlp_desc* desc, void* stream);
/* Split-f16 image of a [128][128] fp32 matrix (rows = output features): two f16 planes (hi, lo) of the matrix scaled by a power
 * of two into the f16 range and stored fragment by fragment (1 KiB per (output block, MFMA step), lane-linear: the image is opaque to
 * the caller), then the inverse scale (csrc/node128s.hip).  `count` matrices in
 * one launch; each image takes nnhip_weight_image_bytes() bytes, 256-byte aligned. */
size_t nnhip_weight_image_bytes(void);
int nnhip_weight_images(const float* const* src, void* const* images, int32_t count, void* stream);
/* the same images in the bf16 format (first plane = the matrix rounded to bf16, unscaled; a format word behind the inverse scale tells
 * the kernels which form an image has): operands of nnhip_mlp_desc.precision = 1 */
int nnhip_weight_images_bf16(const float* const* src, void* const* images, int32_t count, void* stream);
/* how many nnhip_mlp128_ex / nnhip_mlp128_pair_ex calls have taken the bf16 compute mode since the library was loaded */
int64_t nnhip_bf16_mlp_launches(void);
/* two MLPs over the same M rows in one launch (same mode / activation; only the second may accumulate) */
int nnhip_mlp128_pair_ex(const nnhip_mlp_desc* desc0, const nnhip_mlp_desc* desc1, void* stream);

/* ---- tangent kernels (sweeps 3 and 4) ---- */
/* tgeo[e] = (du_e, dx_e): tangent of (dir, x = r/cutoff) along the position direction sign * v[N][3]
 * (training passes v = dL/dF with sign = -1) */
int nnhip_edge_tangent_geom(const float* v, float sign, const int64_t* edge_index, const float* geo, int32_t n_edges,
                            float cutoff, float* tgeo, void* stream);
/* dmsg[p] = deps dx m_i m_j + eps (dm_i m_j + m_i dm_j);  da_mid = da_in + sum_e dmsg   (dm = da_in = NULL: first layer) */
int nnhip_message_tan_fwd(const float* m, const float* dm, const int32_t* xg, const float* tgeo, const float* table,
                          const int32_t* row_ptr, const int32_t* col, const int32_t* pid, const float* da_in, float* dmsg,
                          float* da_mid, int32_t n_atoms, void* stream);
/* df_out = df_in + sum_e dphi1 u + phi1 du + dphi2 f_j + phi2 df_j   (f_in = df_in = NULL: first layer) */
int nnhip_force_message_tan_fwd(const float* phi1, const float* dphi1, const float* phi2, const float* dphi2,
                                const float* geo, const float* tgeo, const int32_t* xg, const int32_t* row_ptr,
                                const int32_t* col, const int32_t* pid, const float* f_in, const float* df_in,
                                float* df_out, int32_t n_atoms, void* stream);
/* dg_h12[p] = (dg_phi1 | dg_phi2),  dg_fin = dgf + sum_e dphi2 gf_j + phi2 dgf_j */
int nnhip_force_message_tan_bwd(const float* gf, const float* dgf, const float* phi2, const float* dphi2, const float* geo,
                                const float* tgeo, const int32_t* xg, const int32_t* row_ptr, const int32_t* col,
                                const int32_t* pid, const float* f_in, const float* df_in, float* dg_h12, float* dg_fin,
                                int32_t n_atoms, void* stream);
/* dg_m[i] = sum_e dG eps m_j + G deps dx m_j + G eps dm_j;  g_eps[p] = G m_i m_j;  dg_eps[p] = dG m_i m_j + G (dm_i m_j + m_i dm_j) */
int nnhip_message_tan_bwd(const float* g_msg, const float* dg_msg, const float* ga, const float* dga, const float* m,
                          const float* dm, const int32_t* xg, const float* tgeo, const float* table,
                          const int32_t* row_ptr, const int32_t* col, const int32_t* pid, float* dg_m, float* g_eps,
                          float* dg_eps, int32_t n_atoms, void* stream);
/* da_out = da_mid + sum_k df_k q_k + f_k dq_k */
int nnhip_update_tan_fwd(const float* da_mid, const float* f, const float* df, const float* q, const float* dq,
                         int32_t n_atoms, float* da_out, void* stream);
/* gq_k = GA f_k;  dgq_k = dGA f_k + GA df_k;  dgf_k = dgf_in,k + dGA q_k + GA dq_k   (the W_u product follows as a linear) */
int nnhip_update_tan_bwd(const float* ga, const float* dga, const float* f, const float* df, const float* q,
                         const float* dq, const float* dgf_in, int32_t n_atoms, float* gq, float* dgq, float* dgf,
                         void* stream);
/* seed of the tangent reverse sweep at the energy head, see csrc/train.hip:head_seed_tan_kernel */
int nnhip_head_seed_tan(const float* e2, const float* de2, const float* w4, const float* b4, const float* scale,
                        const int64_t* z, const int64_t* batch, const float* g_energy, int32_t n_atoms, int32_t activation,
                        float* dg_e2, float* w4row, float* scal, void* stream);
/* rb[p][0:nb] = rbf_e, rb[p][32:32+nb] = drbf_e dx_e for the owner edge of pair p ([P][64], zero padded) */
int nnhip_pair_rbf(const float* rbf, const float* drbf, const float* tgeo, const int64_t* edge_index, const int32_t* pid,
                   int32_t n_edges, int32_t n_basis, float* rb, void* stream);
/* Per-element sums of the first `width` (<= 128) columns of x[N][ldx], two deterministic passes through `scratch`
 * (nnhip_species_scratch_bytes(width)).  Up to two outputs, each a column range of the element table, and a plain total:
 *   out_k[zz][0:cols_k] (pitch ldo_k) = sum_{i: z_i = zz} x[i][c_k : c_k + cols_k];   total[0] = sum_i x[i][c_total]
 * (node_embedding / scale / shift gradients and dL/d b4). */
size_t nnhip_species_scratch_bytes(int32_t width);
int nnhip_species_sum(const float* x, int32_t ldx, int32_t width, const int64_t* z, int32_t n_atoms, float* scratch,
                      float* out0, int32_t c0, int32_t cols0, int32_t ldo0, float* out1, int32_t c1, int32_t cols1,
                      int32_t ldo1, float* total, int32_t c_total, void* stream);

/* Weight gradients dW[o][i] = sum_r A1[r][o] B1[r][i] + A2[r][o] B2[r][i] over M rows, batched; split-K on the fp32 matrix
 * cores with per-workgroup slabs and an ordered final sum.  type 0: operands as stored; 1: B1 = act(hB), B2 = act'(hB) dhB;
 * 2: A2 = A2 * act'(hA).  A2/B2 NULL: one product.  b_cols32: B rows are 32 wide (radial basis).  ld* = 0 means 128.
 * The problem table is DEVICE memory. */
typedef struct {
  const float* A1; const float* B1; const float* A2; const float* B2;
  const float* hA; const float* hB; const float* dhB;
  float* out;
  int32_t M, lda1, lda2, ldb1, ldb2, ldh, type, b_cols32, activation, ldo, ncols, pad_;
} nnhip_wgrad_problem;
size_t nnhip_wgrad_slab_bytes(int32_t n_problems, int32_t chunks);
/* bf16_operands: 0 = fp32 MFMA (v_mfma_f32_32x32x2_f32); 1 = operands rounded to bf16 after their fp32 prologue, fp32
 * accumulation (v_mfma_f32_32x32x16_bf16; torch.autocast(bfloat16)); 2 = fp32-GRADE products from three bf16 pieces per operand
 * (six MFMAs per 16 rows, 24 significant bits, no scaling: the default of fp32 training).
 * pair_rows: the row count of the problems whose M is negative (pair-level problems of a table built for a capacity: the
 * number of pairs changes from batch to batch, the table does not) */
int nnhip_wgrad_batch(const nnhip_wgrad_problem* problems_dev, int32_t n_problems, int32_t chunks, float* slabs,
                      int32_t bf16_operands, int32_t pair_rows, void* stream);
/* Training objective of the reference (newtonnet/train/loss.py:5-50 factory, :53-103 BaseLoss with nn.MSELoss / nn.L1Loss /
 * nn.HuberLoss(delta), mean reduction; scripts/config.yml:45-51) and its gradient:
 *   loss = w[0] sum l_E(E - E*) + w[1] sum l_F(F - F*);  g_energy = w[0] l_E'(E - E*);  g_forces = w[1] l_F'(F - F*)
 * weights_dev[2] = (w_E / n_E, w_F / n_F) lives in DEVICE memory (a data-parallel run refreshes the global counts there).
 * `forces` is whichever force the loss is on (gradient_force, or direct_force for DirectForceLoss, loss.py:41-47). */
enum { NNHIP_LOSS_MSE = 0, NNHIP_LOSS_MAE = 1, NNHIP_LOSS_HUBER = 2 };
int nnhip_loss_grad(const float* energy, const float* energy_label, int32_t n_energy, const float* forces,
                    const float* force_label, int32_t n_force, const float* weights_dev, int32_t mode_energy, int32_t mode_force,
                    float delta_energy, float delta_force, float* loss, float* g_energy, float* g_forces, void* stream);
/* the 'mse' / 'mse' case of nnhip_loss_grad */
int nnhip_mse_loss_grad(const float* energy, const float* energy_label, int32_t n_energy, const float* forces,
                        const float* force_label, int32_t n_force, const float* weights_dev, float* loss, float* g_energy,
                        float* g_forces, void* stream);
/* clip_grad_norm_(max_norm) + torch.optim.Adam step (trainer.py:311-313; no weight decay / amsgrad) on flat buffers of n
 * floats.  state[2] (device) = (step count, last gradient norm); scratch: nnhip_clip_adam_scratch_bytes().  max_norm <= 0:
 * no clipping.
 * nnhip_clip_adam_dev: the hyper-parameters (lr, beta1, beta2, eps, max_norm) are read from DEVICE memory at run time -- a
 * captured HIP graph then follows a learning-rate schedule (trainer.py:190,254: ReduceLROnPlateau on optimizer.param_groups) --
 * and mask_dev (n bytes, optional) marks frozen elements (0 = requires_grad False, newtonnet_train.py:69-81 freeze_*): they
 * stay out of the norm and are not updated. */
size_t nnhip_clip_adam_scratch_bytes(void);
int nnhip_clip_adam(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float* scratch,
                    float* state, float lr, float beta1, float beta2, float eps, float max_norm, void* stream);
int nnhip_clip_adam_dev(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float* scratch,
                        float* state, const float* hyper_dev, const uint8_t* mask_dev, void* stream);
/* out[c] = sum_r src[r][c] for [rows][128] arrays (bias gradients); device table */
typedef struct { const float* src; float* out; int32_t rows; int32_t pad_; } nnhip_colsum_problem;
size_t nnhip_colsum_scratch_bytes(int32_t n);
int nnhip_colsum_batch(const nnhip_colsum_problem* problems_dev, int32_t n, float* scratch, void* stream);

/* --------------------------------------------------------------------------
 * The two halves of a training step as single calls (what newtonnet_amd/train_fused.py:FusedEnergyForces runs in its forward
 * and backward): the stages above in order, on one stream, nothing else.  `ws` holds DEVICE pointers to every buffer of the
 * step (per-layer arrays indexed by layer; sizes as in the stage declarations; the host side owns and sizes them:
 * newtonnet_amd/train_fused.py:TrainWorkspace).  Replaces: the forward + torch.autograd.grad(create_graph=True) of
 * newtonnet/models/newtonnet.py:74-104 / output.py:66-73 (values), and loss.backward() through it, trainer.py:309 (grads).
 *   nnhip_train_values: parameter-only preparation (transposes, filter tables), forward sweep, reverse sweep -> energy[B],
 *                       forces[N][3], every intermediate kept in `ws`.
 *   nnhip_train_grads : given g_energy[B] = dL/dE and g_forces[N][3] = dL/dF: tangent forward, tangent reverse, the batched
 *                       weight-gradient / column-sum / per-element-sum launches; the parameter gradients land where the tables
 *                       (ws->probs, ws->sums) and the g_* pointers say.
 * ------------------------------------------------------------------------ */
typedef struct {
  int32_t n_atoms, n_edges, n_mol, n_layers, n_basis, envelope, bf16_wgrad;
  int32_t flags; /* bit 0: no molecule of the batch has more than NNHIP_MOL_STAGE_MAX atoms (bit 8 of the list's status word is
                    clear): with pair_ptr below, the value sweeps take the molecule-resident force_fwd / msg_bwd forms */
  /* batch + graph (nnhip_graph_* / nnhip_edge_embed outputs; rbf / drbf are required) */
  const int64_t* z; const float* pos; const float* cell; const int64_t* batch;
  const int32_t* mol_ptr; const int32_t* row_ptr; const int32_t* col; const int32_t* rev; const int32_t* pid;
  const int64_t* edge_index; const float* geo; const float* disp; const float* rbf; const float* drbf; const int32_t* xg;
  /* parameter-only data rebuilt by nnhip_train_values */
  float* wT[NNHIP_MAX_LAYERS][7]; float* headT[2]; float* ftab[NNHIP_MAX_LAYERS];
  /* split-f16 images (nnhip_weight_images) of, per layer: update, node0, node2, their transposes, eq1_0, eq1_2, eq2_0, eq2_2,
   * their transposes; of head0, head2 and their transposes */
  void* wimg[NNHIP_MAX_LAYERS][14]; void* himg[4];
  /* values, forward */
  float* a0; float* hn[NNHIP_MAX_LAYERS]; float* m[NNHIP_MAX_LAYERS]; float* msg[NNHIP_MAX_LAYERS];
  float* h1[NNHIP_MAX_LAYERS]; float* h2[NNHIP_MAX_LAYERS]; float* phi1[NNHIP_MAX_LAYERS]; float* phi2[NNHIP_MAX_LAYERS];
  float* a_mid[NNHIP_MAX_LAYERS]; float* a_out[NNHIP_MAX_LAYERS]; float* f_out[NNHIP_MAX_LAYERS]; float* q[NNHIP_MAX_LAYERS];
  float* e1; float* e2; float* g_e2; float* atom_energy; float* energy; float* forces;
  /* values, reverse */
  float* t_e1; float* GA[NNHIP_MAX_LAYERS]; float* gf[NNHIP_MAX_LAYERS]; float* Gf[2]; float* g_h12[NNHIP_MAX_LAYERS];
  float* t1[NNHIP_MAX_LAYERS]; float* t2[NNHIP_MAX_LAYERS]; float* g_msg[NNHIP_MAX_LAYERS]; float* g_m[NNHIP_MAX_LAYERS];
  float* t_n[NNHIP_MAX_LAYERS]; float* g_x; float* g_u; float* g_d;
  /* tangents, forward */
  float* tgeo; float* da_mid; float* da_out[NNHIP_MAX_LAYERS]; float* dhn[NNHIP_MAX_LAYERS]; float* dm[NNHIP_MAX_LAYERS];
  float* dmsg[NNHIP_MAX_LAYERS]; float* dh1[NNHIP_MAX_LAYERS]; float* dh2[NNHIP_MAX_LAYERS]; float* dphi1[NNHIP_MAX_LAYERS];
  float* dphi2[NNHIP_MAX_LAYERS]; float* df_out[NNHIP_MAX_LAYERS]; float* dq[NNHIP_MAX_LAYERS]; float* de1; float* de2;
  /* tangents, reverse */
  float* dg_e2; float* w4row; float* scal; float* dg_e1; float* dGA; float* dgf; float* dGf[2];
  float* gq[NNHIP_MAX_LAYERS]; float* dgq[NNHIP_MAX_LAYERS]; float* dg_h12[NNHIP_MAX_LAYERS]; float* dg_h1[NNHIP_MAX_LAYERS];
  float* dg_h2[NNHIP_MAX_LAYERS]; float* dg_msg; float* g_eps[NNHIP_MAX_LAYERS]; float* dg_eps[NNHIP_MAX_LAYERS];
  float* dg_m[NNHIP_MAX_LAYERS]; float* dg_hn[NNHIP_MAX_LAYERS]; float* rb; const float* zeros_nf;
  /* batched gradient launches and the outputs they do not cover */
  const nnhip_wgrad_problem* probs; const nnhip_colsum_problem* sums; float* slabs; float* cs_scratch; float* sp_scratch;
  int32_t n_probs, chunks, n_sums, pad2_;
  float* g_embedding; float* g_scale; float* g_shift; float* g_head4_b;
  /* layer_norm=True only (NULL otherwise): x_hat [N][F] and 1/sigma [N] of the value sweep, their tangents, the gradient at the
   * LayerNorm OUTPUT kept by the value reverse sweep, and the rows whose column sums are d gamma / d beta */
  float* ln_xhat[NNHIP_MAX_LAYERS]; float* ln_rstd[NNHIP_MAX_LAYERS]; float* ln_dxhat[NNHIP_MAX_LAYERS];
  float* ln_drstd[NNHIP_MAX_LAYERS]; float* ln_gy[NNHIP_MAX_LAYERS]; float* ln_row_w[NNHIP_MAX_LAYERS];
  float* ln_row_b[NNHIP_MAX_LAYERS];
  /* optional (NULL: the row kernels find a row's own pairs by a ballot): pair_ptr[N+1] of nnhip_graph_count_pairs -- the value sweeps
   * then run the forms the inference step runs (round 6) */
  const int32_t* pair_ptr;
} nnhip_train_ws;
size_t nnhip_train_ws_bytes(void); /* sizeof(nnhip_train_ws) of this build (bindings check their mirror against it) */
int nnhip_train_values(const nnhip_model* model, const nnhip_train_ws* ws, void* stream);
int nnhip_train_grads(const nnhip_model* model, const nnhip_train_ws* ws, const float* g_energy, const float* g_forces,
                      void* stream);
/* The same with adjoint seeds at the final node states: seed_a [N][F] = dL/d atom_node, seed_f [N][3][F] = dL/d force_node of
 * loss terms that read them directly (the direct_force head below; either may be NULL).  With g_forces = 0 and no seeds this is
 * plain back-propagation of an energy-only loss (output_properties ['energy'], trainer.py:299-313). */
int nnhip_train_grads_seeded(const nnhip_model* model, const nnhip_train_ws* ws, const float* g_energy, const float* g_forces,
                             const float* seed_a, const float* seed_f, void* stream);
/* First-order adjoint of the direct_force head (output.py:115-132, scalers.py:55-56; DirectForceLoss loss.py:41-47), given
 * g_out [N][3] = dL/d direct_force.  keep: the scratch of the forward nnhip_direct_force call, left untouched by the caller
 * ([3][N][F]: pre-activations of the first two linears, output of the third).  work
 * (nnhip_direct_force_bwd_work_floats(N) floats) receives g_d3 | g_pre2 | t1 | g_pre1 ([N][F] each, in this order): the rows of
 * the head's weight-gradient products  dW4 = g_d3^T act(pre2), dW2 = g_pre2^T act(pre1), dW0 = g_pre1^T atom_node  and of its
 * bias column sums, which the caller's nnhip_wgrad_batch / nnhip_colsum_batch tables reference.  seed_a / seed_f: outputs for
 * nnhip_train_grads_seeded.  g_scale [119] (NULL without a scale): gradient of scalers.k.scale.weight; sc4 [N][4] and
 * sp_scratch (nnhip_species_scratch_bytes(4)): scratch. */
size_t nnhip_direct_force_bwd_work_floats(int32_t n_atoms);
int nnhip_direct_force_bwd(const float* g_out, const float* force_node, const int64_t* z, const float* w0, const float* w2,
                           const float* w4, const float* scale, int32_t activation, int32_t n_atoms, const float* keep,
                           float* work, float* seed_a, float* seed_f, float* sc4, float* sp_scratch, float* g_scale,
                           void* stream);

/* --------------------------------------------------------------------------
 * Product form of the dense kernels.  1 (default): the 128x128 linears of the hot path (edge MLPs, node MLPs, equiv_update
 * and their adjoints / tangents, SiLU models) form each fp32 product from two scaled f16 pieces per operand on
 * v_mfma_f32_32x32x16_f16 with fp32 accumulation (csrc/mlp128s.hip, node128s.hip); 0 (environment NNHIP_MLP_SPLIT=0,
 * read once per process): v_mfma_f32_32x32x2_f32 everywhere.  Inputs, outputs and accumulators are fp32 either way.
 * ------------------------------------------------------------------------ */
int nnhip_split_products(void);
/* Which forms the fused edge-MLP launches of a large batch take (what bench.py prices its byte model with):
 *   bit 0  split-f16 products (nnhip_split_products)
 *   bit 1  the two-MLP adjoint launches run in the one-pass register-weights kernel (mlp_regw_kernel)
 *   bit 2  ... the two-MLP forward launches too
 *   bit 3  the single-MLP adjoint (layer 0) runs in mlp_regw_kernel        bit 4  ... the single-MLP forward too */
int nnhip_mlp_forms(void);
/* Every form choice the library makes in this process -- neighbor-list builder thresholds, waves per receiver row, the
 * molecule-per-workgroup forms and their thresholds, the edge-MLP forms, the fused edge phase -- as one JSON object written into
 * buf (NUL-terminated; NNHIP_E_INVALID when n is too small: 2048 bytes are plenty), with the NNHIP_* environment switches that are
 * set.  For bench lines and for tests that pin non-default forms; the reference has no counterpart (one eager path,
 * newtonnet/models/newtonnet.py:74-104). */
int nnhip_config(char* buf, size_t n);

/* --------------------------------------------------------------------------
 * Internal spatial order of ONE big system (csrc/graph.hip, round 6).  The reference evaluates the atoms in the caller's order
 * (newtonnet/layers/representations.py:74-98 over one molecule); the step of a molecule of >= 16 384 atoms runs on the atoms in Morton
 * order of cells of max(cutoff, extent / 64) and hands everything back in the caller's order (newtonnet_amd/models/newtonnet.py).
 *   nnhip_spatial_order: perm[k] = input index of the atom at position k, inv = its inverse (int32 [N] each), and -- optional -- z / pos
 *     gathered through perm; deterministic (by cell key, then by input index); scratch of nnhip_spatial_order_scratch_bytes(N).
 *   nnhip_permute_rows: out[k] = x[idx[k]] for rows of `width` floats (per-atom results back: idx = inv).
 *   nnhip_edge_index_unpermute: the [2][E] int64 list of the permuted CSR (row_ptr_p, col_p) as the reference lists it for the caller's
 *     order -- rows by ascending i, neighbors by ascending j; scratch of (N + 1 + N / 1024 + 2) ints.
 * ------------------------------------------------------------------------ */
size_t nnhip_spatial_order_scratch_bytes(int32_t n_atoms);
int nnhip_spatial_order(const float* pos, const int64_t* z, int32_t n_atoms, float cutoff, int32_t* perm, int32_t* inv,
                        int64_t* z_out, float* pos_out, void* scratch, void* stream);
int nnhip_permute_rows(const float* x, const int32_t* idx, int32_t n_rows, int32_t width, float* out, void* stream);
int nnhip_edge_index_unpermute(const int32_t* row_ptr_p, const int32_t* col_p, const int32_t* perm, const int32_t* inv,
                               int32_t n_atoms, int32_t n_edges, int64_t* edge_index, int32_t* scratch, void* stream);

/* --------------------------------------------------------------------------
 * Timing hook for bench.py: wraps the kernels of one nnhip_energy_forces call
 * in HIP events on `stream` and accumulates per-kernel-class milliseconds.
 * classes: 0 = edge kernels (message/force fwd+adjoint), 1 = all dense MFMA kernels, 2 = everything else,
 *          3-6 = msg_fwd / force_fwd / force_bwd / msg_bwd, 7 = graph build, 8 = fused edge-MLP kernel (mlp128),
 *          9 = single linears (lin128), 10 = the batched weight-gradient kernel of training (wgrad_kernel, without its
 *          slab reduction), 11 = the one-pass register-weights form of the two edge MLPs (mlp_regw_kernel; its launches are
 *          counted in class 8 as well), 12 / 13 = unused (the molecule-resident fused edge phase of rounds 5-6,
 *          removed: slower than the row kernels, profiles/r06_fused_persistent_ab.txt).  Disabled (0) by default.  on = 1: every class; any other non-zero value is a mask,
 *          bit (k + 1) = class k: only those classes record events (bench.py times each class in a pass of its own, so that a
 *          kernel's duration is not stretched by the events of the kernels around it).
 * ------------------------------------------------------------------------ */
#define NNHIP_N_TIMER_CLASSES 14
int nnhip_timers_enable(int32_t on);
int nnhip_timers_read(double* ms_per_class, int64_t* launches_per_class, int32_t reset);

#ifdef __cplusplus
}
#endif
#endif /* NEWTONNET_HIP_H */
