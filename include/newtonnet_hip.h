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
#define NNHIP_N_ELEMENTS 119 /* rows of node_embedding / scale / shift (z = 0..118) */

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
 *   mol_ptr[B+1].  status is int32[1 + ceil(n_atoms/1024)]: status[0] is set non-zero if `batch` is not sorted, the rest
 *   is scratch of the prefix scan.
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

/* --------------------------------------------------------------------------
 * Verlet-skin reuse of a neighbor list in an MD loop (caller: MLAseCalculator.calculate,
 * newtonnet/utils/ase_interface.py:52-81, which rebuilds the O(N^2) graph every step).
 * nnhip_edge_disp: disp[e] = pos_i - pos_j (+ the reference's single-image shift) for the edges of a list built
 *   earlier with cutoff + skin (edge_index[2][E] int64 from nnhip_graph_fill).  nnhip_edge_embed with the real cutoff
 *   then points candidates with r >= cutoff at all-zero filter rows, so they contribute exactly nothing.
 * ------------------------------------------------------------------------ */
int nnhip_edge_disp(const float* pos, const float* cell, const int64_t* batch, const int64_t* edge_index, int32_t n_edges,
                    float* disp, void* stream);

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
                     float* geo, float* rbf, float* drbf, int32_t* xg, void* stream);

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
  size_t h12[NNHIP_MAX_LAYERS];    /* 2 x [pad32(P)][F] equiv_message{1,2} hidden pre-activations, private to the MLP kernels
                                      (MFMA-fragment order when P > 49152, row-major below) */
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

/* Parameter-only preparation (transposed weights for the reverse sweep, radial-filter tables, layer 0's
 * message_nodepart per element): what the reference gets for free from nn.Module state.  `prepared` is a caller-owned
 * 256-byte-aligned block of nnhip_prepared_bytes(n_layers); fill it with nnhip_prepare whenever the parameters may have
 * changed (the Python mirror does it on every forward, on the stream, while the host waits for the edge count) and pass it
 * to nnhip_energy_forces.  prepared == NULL: nnhip_energy_forces rebuilds the block inside its workspace on every call. */
size_t nnhip_prepared_bytes(int32_t n_layers);
int nnhip_prepare(const nnhip_model* model, void* prepared, size_t prepared_bytes, void* stream);

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

/* --------------------------------------------------------------------------
 * Timing hook for bench.py: wraps the kernels of one nnhip_energy_forces call
 * in HIP events on `stream` and accumulates per-kernel-class milliseconds.
 * classes: 0 = edge kernels (message/force fwd+adjoint), 1 = all dense MFMA kernels, 2 = everything else,
 *          3-6 = msg_fwd / force_fwd / force_bwd / msg_bwd, 7 = graph build, 8 = fused edge-MLP kernel (mlp128),
 *          9 = single linears (lin128).  Disabled (0) by default.
 * ------------------------------------------------------------------------ */
#define NNHIP_N_TIMER_CLASSES 10
int nnhip_timers_enable(int32_t on);
int nnhip_timers_read(double* ms_per_class, int64_t* launches_per_class, int32_t reset);

#ifdef __cplusplus
}
#endif
#endif /* NEWTONNET_HIP_H */
