"""ctypes binding of libnewtonnet_hip.so (C ABI: include/newtonnet_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every kernel is
in the shared library.  There is no fallback: if the library is missing or a
call fails, this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', os.environ.get('NNHIP_LIB_NAME', 'libnewtonnet_hip.so'))  # env: tooling only
BUILD_SCRIPT = os.path.join(_HERE, 'csrc', 'build.sh')

NNHIP_F = 128
NNHIP_NB = 20
NNHIP_MAX_NB = 32
# activation ids of include/newtonnet_hip.h, keyed by the reference's factory names (activations.py:5-30)
ACTIVATION_IDS = {'swish': 0, 'silu': 0, 'relu': 1, 'elu': 2, 'leaky_relu': 3, 'tanh': 4, 'sigmoid': 5, 'softplus': 6,
                  'gelu': 7, 'ssp': 8}
NNHIP_MAX_LAYERS = 8
N_TIMER_CLASSES = 14
TIMER_CLASSES = ('edge_all', 'linear_mfma', 'other', 'edge_msg_fwd', 'edge_force_fwd', 'edge_force_bwd',
                 'edge_msg_bwd', 'graph', 'mlp128', 'lin128', 'wgrad', 'mlp_onepass', 'mol_fwd', 'mol_bwd')

_fp = C.POINTER(C.c_float)


class LayerParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('node0_w', 'node0_b', 'node2_w', 'node2_b', 'edge_w', 'eq1_0_w', 'eq1_2_w',
                                         'eq2_0_w', 'eq2_2_w', 'update_w', 'ln_w', 'ln_b')]


class Model(C.Structure):
    _fields_ = [('n_features', C.c_int32), ('n_basis', C.c_int32), ('n_layers', C.c_int32), ('cutoff', C.c_float),
                ('node_embedding', C.c_void_p), ('frequencies', C.c_void_p),
                ('layer', LayerParams * NNHIP_MAX_LAYERS),
                ('head0_w', C.c_void_p), ('head0_b', C.c_void_p), ('head2_w', C.c_void_p), ('head2_b', C.c_void_p),
                ('head4_w', C.c_void_p), ('head4_b', C.c_void_p), ('scale', C.c_void_p), ('shift', C.c_void_p),
                ('activation', C.c_int32), ('envelope', C.c_int32)]


class WsLayout(C.Structure):
    _fields_ = ([(n, C.c_size_t * NNHIP_MAX_LAYERS) for n in ('m', 'hn', 'msg', 'h12', 'phi1', 'phi2', 'a_mid', 'a_out',
                                                               'f_out', 'q')]
                + [(n, C.c_size_t) for n in ('a0', 'e1', 'e2', 'g_x', 'g_u', 'g_a', 'g_f', 'total')])


class MlpDesc(C.Structure):
    """nnhip_mlp_desc (include/newtonnet_hip.h)."""
    _fields_ = [('X', C.c_void_p), ('ldx', C.c_int32), ('W1', C.c_void_p), ('W2', C.c_void_p), ('b1', C.c_void_p),
                ('b2', C.c_void_p), ('H', C.c_void_p), ('ldh', C.c_int32), ('Y', C.c_void_p), ('ldy', C.c_int32),
                ('M', C.c_int32), ('mode', C.c_int32), ('accumulate', C.c_int32), ('activation', C.c_int32),
                ('T', C.c_void_p), ('T2', C.c_void_p), ('Hd', C.c_void_p), ('G', C.c_void_p),
                ('W1_image', C.c_void_p), ('W2_image', C.c_void_p), ('precision', C.c_int32), ('pad_', C.c_int32)]


class WgradProblem(C.Structure):
    """nnhip_wgrad_problem."""
    _fields_ = [(n, C.c_void_p) for n in ('A1', 'B1', 'A2', 'B2', 'hA', 'hB', 'dhB', 'out')] + \
               [(n, C.c_int32) for n in ('M', 'lda1', 'lda2', 'ldb1', 'ldb2', 'ldh', 'type', 'b_cols32', 'activation', 'ldo',
                                         'ncols', 'pad_')]


class ColsumProblem(C.Structure):
    """nnhip_colsum_problem."""
    _fields_ = [('src', C.c_void_p), ('out', C.c_void_p), ('rows', C.c_int32), ('pad_', C.c_int32)]


def _train_ws_fields():
    L, vp, i32 = NNHIP_MAX_LAYERS, C.c_void_p, C.c_int32
    one = lambda *names: [(n, vp) for n in names]            # noqa: E731
    per = lambda *names: [(n, vp * L) for n in names]        # noqa: E731
    return ([(n, i32) for n in ('n_atoms', 'n_edges', 'n_mol', 'n_layers', 'n_basis', 'envelope', 'bf16_wgrad', 'flags')]
            + one('z', 'pos', 'cell', 'batch', 'mol_ptr', 'row_ptr', 'col', 'rev', 'pid', 'edge_index', 'geo', 'disp', 'rbf',
                  'drbf', 'xg')
            + [('wT', (vp * 7) * L), ('headT', vp * 2)] + per('ftab') + [('wimg', (vp * 14) * L), ('himg', vp * 4)]
            + one('a0') + per('hn', 'm', 'msg', 'h1', 'h2', 'phi1', 'phi2', 'a_mid', 'a_out', 'f_out', 'q')
            + one('e1', 'e2', 'g_e2', 'atom_energy', 'energy', 'forces')
            + one('t_e1') + per('GA', 'gf') + [('Gf', vp * 2)] + per('g_h12', 't1', 't2', 'g_msg', 'g_m', 't_n')
            + one('g_x', 'g_u', 'g_d')
            + one('tgeo', 'da_mid') + per('da_out', 'dhn', 'dm', 'dmsg', 'dh1', 'dh2', 'dphi1', 'dphi2', 'df_out', 'dq')
            + one('de1', 'de2')
            + one('dg_e2', 'w4row', 'scal', 'dg_e1', 'dGA', 'dgf') + [('dGf', vp * 2)]
            + per('gq', 'dgq', 'dg_h12', 'dg_h1', 'dg_h2') + one('dg_msg') + per('g_eps', 'dg_eps', 'dg_m', 'dg_hn')
            + one('rb', 'zeros_nf')
            + one('probs', 'sums', 'slabs', 'cs_scratch', 'sp_scratch')
            + [(n, i32) for n in ('n_probs', 'chunks', 'n_sums', 'pad2_')]
            + one('g_embedding', 'g_scale', 'g_shift', 'g_head4_b')
            + per('ln_xhat', 'ln_rstd', 'ln_dxhat', 'ln_drstd', 'ln_gy', 'ln_row_w', 'ln_row_b')
            + one('pair_ptr'))


class TrainWs(C.Structure):
    """nnhip_train_ws: device pointers of one training step (newtonnet_amd/train_fused.py:TrainWorkspace fills it)."""
    _fields_ = _train_ws_fields()


MODE_FWD, MODE_BWD, MODE_TAN, MODE_TAN2 = 0, 1, 2, 3
LOSS_MODES = {'mse': 0, 'mae': 1, 'huber': 2}          # newtonnet/train/loss.py:53-103
WG_PLAIN, WG_ACT, WG_TDACT = 0, 1, 2


class HipLibraryError(RuntimeError):
    pass


_lib = None


def build(verbose: bool = False, force: bool = False) -> str:
    """Compile the HIP sources for gfx950 into newtonnet_amd/lib/libnewtonnet_hip.so (force: every source, not only the
    stale ones)."""
    r = subprocess.run(['bash', BUILD_SCRIPT] + (['--force'] if force else []), capture_output=True, text=True)
    if r.returncode != 0:
        raise HipLibraryError(f'hipcc build failed:\n{r.stdout}\n{r.stderr}')
    if verbose:
        print(r.stdout.strip())
    return LIB_PATH


def lib():
    """The loaded library.  Fails loudly when it is absent (no CPU path exists in this package)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            f'or `bash {BUILD_SCRIPT}`.  newtonnet_amd has no CPU fallback.')
    L = C.CDLL(LIB_PATH)
    vp, i32, f32, sz = C.c_void_p, C.c_int32, C.c_float, C.c_size_t
    L.nnhip_version.restype = C.c_int
    L.nnhip_build_flags.restype = C.c_int
    if L.nnhip_build_flags() & 1 and os.environ.get('NNHIP_ALLOW_TOOLING_LIB') != '1':
        raise HipLibraryError(f'{LIB_PATH} is a TOOLING build (compiled with extra flags such as an ablation switch: its results '
                              'may be wrong).  Rebuild with `bash newtonnet_amd/csrc/build.sh --force`, or set '
                              'NNHIP_ALLOW_TOOLING_LIB=1 for measurements.')
    L.nnhip_last_error.restype = C.c_char_p
    L.nnhip_graph_count.argtypes = [vp, vp, vp, i32, i32, f32, vp, vp, vp, vp]
    L.nnhip_graph_fill.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, f32, vp, vp, vp, vp, vp]
    L.nnhip_graph_count_pairs.argtypes = [vp, vp, vp, i32, i32, f32, vp, vp, vp, vp, vp]
    L.nnhip_graph_pair_scan.argtypes = [vp, i32, vp, vp]
    L.nnhip_graph_count_cells_pairs.argtypes = [vp, vp, i32, f32, _fp, vp, vp, vp, vp, vp]
    L.nnhip_graph_finish_cells.argtypes = [vp, vp, i32, i32, f32, _fp] + [vp] * 9 + [i32, vp, vp, vp, vp, i32, vp]
    L.nnhip_graph_finish.argtypes = [vp] * 6 + [i32, i32, i32, f32] + [vp] * 6 + [i32, vp, vp, vp, vp, i32, vp]
    L.nnhip_graph_finish_early.argtypes = L.nnhip_graph_finish.argtypes
    L.nnhip_graph_finish_dev.argtypes = L.nnhip_graph_finish.argtypes[:-1] + [vp, vp, vp, i32, vp]
    L.nnhip_energy_forces_dev.argtypes = [C.POINTER(Model), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, sz,
                                          vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.nnhip_mlp_forms.restype = C.c_int
    L.nnhip_step_layout_of.argtypes = [i32, i32, i32, vp]
    L.nnhip_step_layout_of.restype = C.c_int
    L.nnhip_forward_dev.argtypes = [C.POINTER(Model), vp, vp]
    L.nnhip_forward_dev.restype = C.c_int
    L.nnhip_edge_embed.argtypes = [vp, i32, f32, vp, i32, vp, vp, vp, vp, i32, vp]
    L.nnhip_edge_disp.argtypes = [vp, vp, vp, vp, i32, vp, vp]
    L.nnhip_edge_refresh.argtypes = [vp, vp, vp, vp, i32, f32, vp, i32, vp, vp, vp, vp, vp, i32, vp]
    L.nnhip_check_species.argtypes = [vp, i32, vp, vp]
    L.nnhip_graph_cells_scratch_bytes.argtypes = [i32, _fp, f32]
    L.nnhip_graph_cells_scratch_bytes.restype = sz
    L.nnhip_graph_count_cells.argtypes = [vp, vp, i32, f32, _fp, vp, vp, vp, vp]
    L.nnhip_graph_fill_cells.argtypes = [vp, vp, i32, i32, f32, _fp, vp, vp, vp, vp, vp, vp, vp]
    L.nnhip_workspace_bytes.argtypes = [i32, i32, i32, i32]
    L.nnhip_workspace_bytes.restype = sz
    L.nnhip_workspace_layout.argtypes = [i32, i32, i32, i32, C.POINTER(WsLayout)]
    L.nnhip_graph_pairs.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp, vp]
    L.nnhip_energy_forces.argtypes = [C.POINTER(Model), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, sz,
                                      vp, vp, vp, vp, vp, vp, vp, vp]
    L.nnhip_energy_forces_pp.argtypes = L.nnhip_energy_forces.argtypes[:-1] + [vp, i32, vp]
    L.nnhip_energy_forces_pp.restype = C.c_int
    L.nnhip_edge_index_from_csr.argtypes = [vp, vp, i32, i32, vp, vp, vp]
    L.nnhip_prepared_bytes.argtypes = [i32]
    L.nnhip_prepared_bytes.restype = sz
    L.nnhip_prepare.argtypes = [C.POINTER(Model), vp, sz, vp]
    L.nnhip_prepare_check.argtypes = [C.POINTER(Model), vp, sz, vp, i32, vp]
    L.nnhip_linear128.argtypes = [vp, i32, vp, vp, i32, vp, vp, i32, i32, i32, i32, vp]
    L.nnhip_mlp128.argtypes = [vp, i32, vp, vp, vp, i32, vp, i32, i32, i32, i32, vp]
    L.nnhip_direct_force.argtypes = [vp] * 10 + [i32, i32, vp, vp, vp]
    L.nnhip_segment_sum.argtypes = [vp, vp, i32, i32, vp, vp]
    L.nnhip_gather_rows.argtypes = [vp, vp, i32, i32, vp, vp]
    L.nnhip_timers_enable.argtypes = [i32]
    L.nnhip_timers_read.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_int64), i32]
    # per-stage entry points and training kernels
    pp = C.POINTER(vp)
    L.nnhip_embed.argtypes = [vp, vp, i32, vp, vp]
    L.nnhip_filter_table_bytes.restype = sz
    L.nnhip_filter_tables.argtypes = [pp, pp, i32, vp, i32, i32, vp]
    L.nnhip_transpose128.argtypes = [pp, pp, i32, vp]
    L.nnhip_message_fwd.argtypes = [vp] * 9 + [i32, vp]
    L.nnhip_message_bwd.argtypes = [vp] * 10 + [i32, i32, vp]
    L.nnhip_force_message_fwd.argtypes = [vp] * 9 + [i32, vp]
    L.nnhip_force_message_bwd.argtypes = [vp] * 12 + [i32, vp]
    L.nnhip_edge_embed_bwd.argtypes = [vp] * 10 + [i32, i32, i32, i32, f32, vp, vp, vp, vp]
    L.nnhip_node_fwd.argtypes = [vp] * 11 + [i32, i32, vp]
    L.nnhip_node_bwd.argtypes = [vp] * 5 + [i32] + [vp] * 5 + [i32, i32, vp]
    L.nnhip_head_out.argtypes = [vp] * 7 + [i32, i32, i32, vp, vp, vp, vp]
    L.nnhip_mlp128_ex.argtypes = [C.POINTER(MlpDesc), vp]
    L.nnhip_mlp128_pair_ex.argtypes = [C.POINTER(MlpDesc), C.POINTER(MlpDesc), vp]
    L.nnhip_edge_tangent_geom.argtypes = [vp, f32, vp, vp, i32, f32, vp, vp]
    L.nnhip_message_tan_fwd.argtypes = [vp] * 11 + [i32, vp]
    L.nnhip_force_message_tan_fwd.argtypes = [vp] * 13 + [i32, vp]
    L.nnhip_force_message_tan_bwd.argtypes = [vp] * 14 + [i32, vp]
    L.nnhip_message_tan_bwd.argtypes = [vp] * 15 + [i32, vp]
    L.nnhip_update_tan_fwd.argtypes = [vp] * 5 + [i32, vp, vp]
    L.nnhip_update_tan_bwd.argtypes = [vp] * 7 + [i32, vp, vp, vp, vp]
    L.nnhip_head_seed_tan.argtypes = [vp] * 8 + [i32, i32, vp, vp, vp, vp]
    L.nnhip_pair_rbf.argtypes = [vp] * 5 + [i32, i32, vp, vp]
    L.nnhip_species_scratch_bytes.argtypes = [i32]
    L.nnhip_species_scratch_bytes.restype = sz
    L.nnhip_species_sum.argtypes = [vp, i32, i32, vp, i32, vp, vp, i32, i32, i32, vp, i32, i32, i32, vp, i32, vp]
    L.nnhip_colsum_scratch_bytes.argtypes = [i32]
    L.nnhip_colsum_scratch_bytes.restype = sz
    L.nnhip_wgrad_slab_bytes.argtypes = [i32, i32]
    L.nnhip_wgrad_slab_bytes.restype = sz
    L.nnhip_wgrad_batch.argtypes = [vp, i32, i32, vp, i32, i32, vp]
    L.nnhip_colsum_batch.argtypes = [vp, i32, vp, vp]
    L.nnhip_train_values.argtypes = [C.POINTER(Model), C.POINTER(TrainWs), vp]
    L.nnhip_train_grads.argtypes = [C.POINTER(Model), C.POINTER(TrainWs), vp, vp, vp]
    L.nnhip_train_grads_seeded.argtypes = [C.POINTER(Model), C.POINTER(TrainWs), vp, vp, vp, vp, vp]
    L.nnhip_direct_force_bwd_work_floats.argtypes = [i32]
    L.nnhip_direct_force_bwd_work_floats.restype = sz
    L.nnhip_direct_force_bwd.argtypes = [vp] * 7 + [i32, i32] + [vp] * 8
    L.nnhip_train_ws_bytes.restype = sz
    L.nnhip_weight_image_bytes.restype = sz
    L.nnhip_weight_images.argtypes = [vp, vp, i32, vp]
    L.nnhip_mse_loss_grad.argtypes = [vp, vp, i32, vp, vp, i32, vp, vp, vp, vp, vp]
    L.nnhip_loss_grad.argtypes = [vp, vp, i32, vp, vp, i32, vp, i32, i32, f32, f32, vp, vp, vp, vp]
    L.nnhip_clip_adam_dev.argtypes = [vp, vp, vp, vp, C.c_int64, vp, vp, vp, vp, vp]
    L.nnhip_clip_adam_scratch_bytes.restype = sz
    L.nnhip_clip_adam.argtypes = [vp, vp, vp, vp, C.c_int64, vp, vp, f32, f32, f32, f32, f32, vp]
    for fn in STAGE_SYMBOLS:
        if fn not in ('nnhip_filter_table_bytes', 'nnhip_wgrad_slab_bytes', 'nnhip_species_scratch_bytes',
                      'nnhip_colsum_scratch_bytes', 'nnhip_clip_adam_scratch_bytes', 'nnhip_train_ws_bytes',
                      'nnhip_weight_image_bytes', 'nnhip_direct_force_bwd_work_floats'):
            getattr(L, fn).restype = C.c_int
    for fn in ('nnhip_graph_count', 'nnhip_graph_fill', 'nnhip_edge_embed', 'nnhip_workspace_layout',
               'nnhip_energy_forces', 'nnhip_timers_enable', 'nnhip_timers_read', 'nnhip_linear128', 'nnhip_segment_sum', 'nnhip_gather_rows', 'nnhip_graph_count_cells',
               'nnhip_graph_fill_cells', 'nnhip_mlp128', 'nnhip_graph_pairs', 'nnhip_direct_force', 'nnhip_edge_disp',
               'nnhip_prepare', 'nnhip_prepare_check', 'nnhip_check_species', 'nnhip_graph_count_pairs', 'nnhip_graph_pair_scan', 'nnhip_graph_finish', 'nnhip_graph_finish_early', 'nnhip_edge_refresh', 'nnhip_graph_count_cells_pairs', 'nnhip_graph_finish_cells',
               'nnhip_graph_finish_dev', 'nnhip_energy_forces_dev'):
        getattr(L, fn).restype = C.c_int
    _lib = L
    return L


STAGE_SYMBOLS = ('nnhip_embed', 'nnhip_filter_table_bytes', 'nnhip_filter_tables', 'nnhip_transpose128', 'nnhip_message_fwd',
                 'nnhip_message_bwd', 'nnhip_force_message_fwd', 'nnhip_force_message_bwd', 'nnhip_edge_embed_bwd',
                 'nnhip_node_fwd', 'nnhip_node_bwd', 'nnhip_head_out', 'nnhip_mlp128_ex', 'nnhip_mlp128_pair_ex', 'nnhip_edge_tangent_geom',
                 'nnhip_message_tan_fwd', 'nnhip_force_message_tan_fwd', 'nnhip_force_message_tan_bwd',
                 'nnhip_message_tan_bwd', 'nnhip_update_tan_fwd', 'nnhip_update_tan_bwd', 'nnhip_head_seed_tan',
                 'nnhip_pair_rbf', 'nnhip_species_sum', 'nnhip_species_scratch_bytes', 'nnhip_wgrad_slab_bytes',
                 'nnhip_wgrad_batch', 'nnhip_colsum_batch', 'nnhip_colsum_scratch_bytes', 'nnhip_mse_loss_grad', 'nnhip_loss_grad',
                 'nnhip_clip_adam', 'nnhip_clip_adam_dev', 'nnhip_clip_adam_scratch_bytes', 'nnhip_train_values', 'nnhip_train_grads',
                 'nnhip_train_ws_bytes', 'nnhip_weight_image_bytes', 'nnhip_weight_images', 'nnhip_train_grads_seeded',
                 'nnhip_direct_force_bwd', 'nnhip_direct_force_bwd_work_floats')

EXPORTED_SYMBOLS = STAGE_SYMBOLS + ('nnhip_version', 'nnhip_last_error', 'nnhip_graph_count', 'nnhip_graph_fill', 'nnhip_edge_embed',
                    'nnhip_workspace_bytes', 'nnhip_workspace_layout', 'nnhip_energy_forces', 'nnhip_timers_enable',
                    'nnhip_timers_read', 'nnhip_linear128', 'nnhip_segment_sum', 'nnhip_gather_rows',
                    'nnhip_graph_cells_scratch_bytes', 'nnhip_graph_count_cells', 'nnhip_graph_fill_cells',
                    'nnhip_mlp128', 'nnhip_graph_pairs', 'nnhip_direct_force', 'nnhip_edge_disp', 'nnhip_prepared_bytes',
                    'nnhip_prepare', 'nnhip_prepare_check', 'nnhip_check_species', 'nnhip_split_products', 'nnhip_build_flags', 'nnhip_graph_count_pairs',
                    'nnhip_graph_pair_scan', 'nnhip_graph_finish', 'nnhip_graph_finish_early', 'nnhip_edge_refresh', 'nnhip_graph_count_cells_pairs',
                    'nnhip_graph_finish_cells', 'nnhip_graph_finish_dev', 'nnhip_energy_forces_dev', 'nnhip_mlp_forms',
                    'nnhip_step_layout_of', 'nnhip_forward_dev', 'nnhip_graph_small_dev', 'nnhip_graph_small_max_atoms',
                    'nnhip_energy_forces_pp', 'nnhip_graph_count_pairs_z', 'nnhip_prepare_check_counter', 'nnhip_graph_mol_dev',
                    'nnhip_edge_index_from_csr', 'nnhip_config', 'nnhip_weight_images_bf16', 'nnhip_bf16_mlp_launches',
                    'nnhip_spatial_order_scratch_bytes', 'nnhip_spatial_order', 'nnhip_permute_rows', 'nnhip_edge_index_unpermute')


def _check(rc: int, what: str):
    if rc != 0:
        raise HipLibraryError(f'{what} failed (code {rc}): {lib().nnhip_last_error().decode()}')


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise NotImplementedError(f'{name}: the HIP path computes in float32 (got {t.dtype})')
    return t.contiguous()


CELL_LIST_MIN_ATOMS = 2048   # below this the all-pairs kernel is at least as fast


_EDGE_FIELDS = ('col', 'rev', 'disp', 'edge_index', 'geo', 'rbf', 'drbf', 'xg', 'pid')


class Graph:
    """Neighbor list + edge embedding of one batch (device tensors).  The per-edge arrays (`col`, `rev`, `pid`, `xg`, `geo`,
    `disp`, `rbf`, `drbf`, `edge_index`) are views into three allocations; they are created on first access -- the host time
    between the edge-count read-back and the first launch behind it is on the step's critical path, and a dozen view
    constructions are ~15 us of it -- and `edge_ptr(name)` gives the raw device address without creating one."""
    __slots__ = ('n_atoms', 'n_mol', 'n_edges', 'mol_ptr', 'row_ptr', 'col', 'rev', 'disp', 'edge_index', 'geo',
                 'rbf', 'drbf', 'xg', 'pid', 'pair_ptr', '_train_eg', 'envelope', 'status', '_arrays', '_cap', '_nb', '_want_rbf',
                 '_meta')

    def __getattr__(self, name):   # (only reached when the slot is still empty)
        if name in _EDGE_FIELDS and self._bind():
            return object.__getattribute__(self, name)
        raise AttributeError(name)

    def _bind(self) -> bool:
        try:
            ints, flts, ei = object.__getattribute__(self, '_arrays')
        except AttributeError:
            return False
        cap, E, nb = self._cap, self.n_edges, self._nb
        self.xg, self.col = ints[:2 * cap].view(cap, 2)[:E], ints[2 * cap:2 * cap + E]
        self.rev, self.pid = ints[3 * cap:3 * cap + E], ints[4 * cap:4 * cap + E]
        self.geo, self.disp = flts[:4 * cap].view(cap, 4)[:E], flts[4 * cap:7 * cap].view(cap, 3)[:E]
        want_rbf = self._want_rbf
        self.rbf = flts[7 * cap:(7 + nb) * cap].view(cap, nb)[:E] if want_rbf else None     # dist_edge (tests / API)
        self.drbf = flts[(7 + nb) * cap:].view(cap, nb)[:E] if want_rbf else None
        self.edge_index = ei[:2 * E].view(2, E) if ei is not None else None   # (its rows were written at stride E)
        return True

    def edge_ptr(self, name: str):
        """Device address of a per-edge array (None when absent), without materialising the view."""
        try:
            ints, flts, ei = object.__getattribute__(self, '_arrays')
        except AttributeError:
            return _ptr(getattr(self, name))
        cap, nb = self._cap, self._nb
        if cap == 0:
            return None
        off = {'xg': (ints, 0), 'col': (ints, 8 * cap), 'rev': (ints, 12 * cap), 'pid': (ints, 16 * cap), 'geo': (flts, 0),
               'disp': (flts, 16 * cap)}
        if name in off:
            base, o = off[name]
            return C.c_void_p(base.data_ptr() + o)
        if name == 'edge_index':
            return _ptr(ei)
        if not self._want_rbf:
            return None
        return C.c_void_p(flts.data_ptr() + (28 * cap if name == 'rbf' else (28 + 4 * nb) * cap))


def prepare(model: Model, device, block: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Fill a prepared block (nnhip_prepare) for `model` on the current stream and return it (a new one unless `block`)."""
    L = lib()
    n = L.nnhip_prepared_bytes(model.n_layers)
    buf = block if block is not None else torch.empty(max(n, 256), dtype=torch.uint8, device=device)
    _check(L.nnhip_prepare(C.byref(model), _ptr(buf), buf.numel(), _stream(buf.device)), 'nnhip_prepare')
    return buf


STATUS_BIG_MOLECULE = 8     # bit of the graph status word: a molecule of more than NNHIP_MOL_STAGE_MAX atoms (informational)
STATUS_PARAMS_CHANGED = 4   # bit of the graph status word: nnhip_prepare_check found a parameter that differs from its snapshot


def prepare_check(model: Model, block: torch.Tensor, status: torch.Tensor):
    """Queue the exact parameters-vs-snapshot comparison of `block` (nnhip_prepare_check); STATUS_PARAMS_CHANGED is OR-ed into
    the device int32 `status[0]` when a parameter changed since the previous check of this block."""
    _check(lib().nnhip_prepare_check(C.byref(model), _ptr(block), block.numel(), _ptr(status), STATUS_PARAMS_CHANGED,
                                     _stream(block.device)), 'nnhip_prepare_check')


def _orthorhombic_box(cell: torch.Tensor, cutoff: float, cell_host=None):
    """Box lengths (3 floats) when `cell` ([1,3,3]) is an axis-aligned periodic box of at least 3 cutoffs per side -- the
    precondition of the O(N) cell-list kernels -- else None.  The CURRENT contents of the cell decide, every call: a device
    tensor is read back (36 bytes, one round trip of ~30 us on a step that takes milliseconds at the >= 2048-atom sizes this
    path serves); callers that hold the cell on the host (the ASE calculator) pass it in and skip the round trip.  Nothing is
    cached on tensor identity: a fresh cell tensor per NPT frame reuses the allocator's block and its version counter."""
    if cell_host is not None:
        c = torch.as_tensor(cell_host, dtype=torch.float32).reshape(3, 3)
    else:
        c = cell.reshape(3, 3).cpu()
    diag = torch.diagonal(c)
    ok = bool((c - torch.diag(diag) == 0).all()) and bool((diag >= 3.0003 * cutoff).all())
    return tuple(float(v) for v in diag) if ok else None


def build_graph(pos: torch.Tensor, cell: torch.Tensor, batch: torch.Tensor, cutoff: float,
                frequencies: torch.Tensor, want_edge_index: bool = True, want_rbf: bool = False,
                while_waiting=None, z: Optional[torch.Tensor] = None, cell_host=None, envelope: int = 9,
                before_sync=None, edge_capacity: int = 0) -> Graph:
    """RadiusGraph + ScaledNorm + envelope x Bessel (representations.py:20-43) on the GPU.
    `while_waiting`: callable run after the counting kernels are queued and before the host waits for the edge count --
    work it launches on the stream fills the GPU's idle time during that round trip (NewtonNet.forward passes
    nnhip_prepare here).  `z` (int64, optional): species, range-checked on the device in the same round trip (the
    reference raises IndexError for z outside 0..118).  `cell_host`: the cell as a host array, when the caller has it.
    `before_sync(status)`: callable that may queue kernels OR-ing further bits (>= 4) into the device status word before it is
    read back with the edge count; the word comes back as `graph.status`.
    `edge_capacity` > 0 (with `while_waiting`; all-pairs builder): the edge arrays are allocated for that many edges and
    nnhip_graph_finish_early is queued BEFORE the host waits for the count -- the kernels read it on the device; when the
    count turns out larger than the capacity they have written nothing and the ordinary path runs after the wait."""
    L = lib()
    dev = pos.device
    pos = _f32c(pos, 'pos')
    cell = _f32c(cell, 'cell')
    batch = batch.contiguous()
    if batch.dtype != torch.int64:
        batch = batch.long()
    N, B = pos.shape[0], cell.shape[0]
    g = Graph()
    g.n_atoms, g.n_mol, g.envelope = N, B, int(envelope)
    n_scan = (N + 1023) // 1024 + 1
    # one int32 block: mol_ptr [B+1] | row_ptr [N+1] | status + scan scratch [1 + n_scan] | pair_ptr [N+1] | its scan scratch
    meta = torch.empty(B + 1 + N + 1 + 1 + n_scan + N + 1 + n_scan, dtype=torch.int32, device=dev)
    g.mol_ptr, g.row_ptr, status = meta[:B + 1], meta[B + 1:B + N + 2], meta[B + N + 2:B + N + 3 + n_scan]
    o_pp = B + N + 3 + n_scan
    g.pair_ptr, pair_scan = meta[o_pp:o_pp + N + 1], meta[o_pp + N + 1:]
    st = _stream(dev)
    # One big orthorhombic periodic box -> O(N) cell-list kernels (bit-identical output to the all-pairs kernels).
    box = None
    if B == 1 and N >= CELL_LIST_MIN_ATOMS:
        lengths = _orthorhombic_box(cell, cutoff, cell_host)
        if lengths is not None:
            box = (C.c_float * 3)(*lengths)
    if box is not None:
        scratch = torch.empty(L.nnhip_graph_cells_scratch_bytes(N, box, float(cutoff)), dtype=torch.uint8, device=dev)
        status[:1].zero_()
        _check(L.nnhip_graph_count_cells_pairs(_ptr(pos), _ptr(cell), N, float(cutoff), box, _ptr(scratch), _ptr(g.mol_ptr),
                                               _ptr(g.row_ptr), _ptr(g.pair_ptr), st), 'nnhip_graph_count_cells_pairs')
    else:   # (the count pass also takes the per-row pair counts: the pair ids then need no pass of their own after the sync)
        _check(L.nnhip_graph_count_pairs(_ptr(pos), _ptr(cell), _ptr(batch), N, B, float(cutoff), _ptr(g.mol_ptr),
                                         _ptr(g.row_ptr), _ptr(status), _ptr(g.pair_ptr), st), 'nnhip_graph_count_pairs')
    if z is not None:
        if z.dtype != torch.int64 or not z.is_contiguous():
            z = z.long().contiguous()
        _check(L.nnhip_check_species(_ptr(z), N, _ptr(status), st), 'nnhip_check_species')
    if before_sync is not None:
        before_sync(status[:1])
    tail_dev = meta[B + N + 1:B + N + 3]
    nb = frequencies.numel()
    freq = _f32c(frequencies, 'frequencies')

    def edge_arrays(cap):
        # everything sized by the edge count in three allocations (the host time between the sync and the first launch is on the
        # step's critical path): int32 [xg 2 cap | col | rev | pid], float32 [geo 4 cap | disp 3 cap | rbf, drbf nb cap each], edge_index
        ints = torch.empty(5 * cap, dtype=torch.int32, device=dev)
        flts = torch.empty((7 + (2 * nb if want_rbf else 0)) * cap, dtype=torch.float32, device=dev)
        ei = torch.empty(2 * cap, dtype=torch.int64, device=dev) if want_edge_index else None
        return ints, flts, ei

    def bind(arrays, cap, E):   # (views of the first E edges: made on first access, Graph._bind)
        g._arrays, g._cap, g._nb, g._want_rbf = arrays, cap, nb, bool(want_rbf)

    def finish_args(arrays, cap):
        ints, flts, ei = arrays
        return (_ptr(pos), _ptr(cell), _ptr(batch), _ptr(g.mol_ptr), _ptr(g.row_ptr), _ptr(g.pair_ptr), N, B, cap, float(cutoff),
                C.c_void_p(ints.data_ptr() + 8 * cap), C.c_void_p(ints.data_ptr() + 12 * cap), C.c_void_p(ints.data_ptr() + 16 * cap),
                C.c_void_p(flts.data_ptr() + 16 * cap), _ptr(ei), _ptr(freq), nb, _ptr(flts),
                C.c_void_p(flts.data_ptr() + 28 * cap) if want_rbf else None,
                C.c_void_p(flts.data_ptr() + (28 + 4 * nb) * cap) if want_rbf else None, _ptr(ints), g.envelope, st)

    early = None
    if while_waiting is not None:
        tail_host = torch.empty(2, dtype=torch.int32, pin_memory=True)
        tail_host.copy_(tail_dev, non_blocking=True)          # queue the read-back first ...
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        # ... then work that does not need the edge count on the host: the pair scan,
        _check(L.nnhip_graph_pair_scan(_ptr(g.pair_ptr), N, _ptr(pair_scan), st), 'nnhip_graph_pair_scan')
        if edge_capacity > 0 and box is None:                  # the fill itself, into arrays of the caller's capacity,
            early = edge_arrays(int(edge_capacity))
            _check(L.nnhip_graph_finish_early(*finish_args(early, int(edge_capacity))), 'nnhip_graph_finish_early')
        while_waiting()                                        # the caller's (allocations), then wait for the copy only
        ev.synchronize()
        tail = tail_host.tolist()
    else:
        _check(L.nnhip_graph_pair_scan(_ptr(g.pair_ptr), N, _ptr(pair_scan), st), 'nnhip_graph_pair_scan')
        tail = tail_dev.tolist()  # (E, status): the one device->host sync of the path
    E, bad = int(tail[0]), int(tail[1])
    if bad & 1:
        raise ValueError('batch must be non-decreasing with values in [0, cell.shape[0]) (PyG collation order)')
    if bad & 2:
        raise IndexError('atomic numbers z must lie in [0, 118] (rows of node_embedding / scale / shift)')
    g.n_edges = E
    g.status = bad | (STATUS_BIG_MOLECULE if box is not None else 0)   # (the cell-list builder serves ONE large system)
    if early is not None and E <= int(edge_capacity):          # the early launch has done the work
        bind(early, int(edge_capacity), E)
        if E == 0:
            g.pair_ptr.zero_()
        return g
    arrays = edge_arrays(E)
    bind(arrays, E, E)
    if box is None:
        _check(L.nnhip_graph_finish(*finish_args(arrays, E)), 'nnhip_graph_finish')
        if E == 0:
            g.pair_ptr.zero_()
    else:
        _check(L.nnhip_graph_finish_cells(_ptr(pos), _ptr(cell), N, E, float(cutoff), box, _ptr(scratch), _ptr(g.row_ptr),
                                          _ptr(g.pair_ptr), _ptr(g.col), _ptr(g.rev), _ptr(g.pid), _ptr(g.disp), _ptr(g.edge_index),
                                          _ptr(freq), nb, _ptr(g.geo), _ptr(g.rbf), _ptr(g.drbf), _ptr(g.xg), g.envelope, st),
               'nnhip_graph_finish_cells')
    return g


def refresh_graph(g: Graph, pos: torch.Tensor, cell: torch.Tensor, batch: torch.Tensor, cutoff: float,
                  frequencies: torch.Tensor) -> Graph:
    """Re-evaluate the geometry of an existing (candidate) list at new positions, in place and without any host sync:
    disp, geo and the filter-table positions xg.  `g` must have been built with want_edge_index=True and a cutoff of at
    least `cutoff` (cutoff + skin for Verlet reuse); candidates outside `cutoff` get all-zero filter rows."""
    L = lib()
    if g.edge_index is None:
        raise ValueError('refresh_graph needs a graph built with want_edge_index=True')
    st = _stream(g.row_ptr.device)     # (pos may be a pinned HOST array read in place by the kernels: the MD loop's zero-copy input)
    pos, cell = _f32c(pos, 'pos'), _f32c(cell, 'cell')
    if batch.dtype != torch.int64 or not batch.is_contiguous():
        batch = batch.long().contiguous()
    E = g.n_edges
    _check(L.nnhip_edge_refresh(_ptr(pos), _ptr(cell), _ptr(batch), _ptr(g.edge_index), E, float(cutoff),
                                _ptr(_f32c(frequencies, 'frequencies')), frequencies.numel(), _ptr(g.disp), _ptr(g.geo), _ptr(g.rbf),
                                _ptr(g.drbf), _ptr(g.xg), g.envelope, st), 'nnhip_edge_refresh')
    return g


class StepLayout(C.Structure):
    """nnhip_step_layout."""
    _fields_ = [(n, C.c_size_t) for n in ('i32_count', 'f32_count', 'mol_ptr', 'row_ptr', 'status', 'pair_ptr', 'pair_scan',
                                          'tail', 'mol_scratch', 'xg', 'col', 'rev', 'pid', 'geo', 'disp', 'energy', 'forces', 'virial',
                                          'atom_energy')]


class StepDev(C.Structure):
    """nnhip_step_dev."""
    _fields_ = ([(n, C.c_void_p) for n in ('z', 'pos', 'cell', 'batch')]
                + [(n, C.c_int32) for n in ('n_atoms', 'n_mol', 'capacity', 'want_forces', 'want_virial', 'seq', 'flags', 'pad_')]
                + [(n, C.c_void_p) for n in ('i32', 'f32', 'edge_index', 'atom_node', 'force_node', 'workspace')]
                + [('workspace_bytes', C.c_size_t), ('prepared', C.c_void_p), ('prepared_bytes', C.c_size_t),
                   ('tail_host', C.c_void_p), ('event', C.c_void_p)])


_step_layouts = {}


def step_layout(N: int, B: int, cap: int) -> StepLayout:
    key = (N, B, cap)
    lay = _step_layouts.get(key)
    if lay is None:
        lay = StepLayout()
        _check(lib().nnhip_step_layout_of(N, B, cap, C.byref(lay)), 'nnhip_step_layout_of')
        if len(_step_layouts) > 256:
            _step_layouts.clear()
        _step_layouts[key] = lay
    return lay


class DevStep:
    """Arenas of one deferred step (nnhip_forward_dev) and lazily made views of what a caller may look at.  Quacks like the
    result dict of energy_forces (`step[name]`) and like a Graph as far as NewtonNet.forward needs one (`edge_index`,
    `n_edges`, `status`).  `n_edges` is None until the caller has read the count."""
    __slots__ = ('N', 'B', 'cap', 'lay', 'i32', 'f32', 'ei', 'atom_node', 'force_node', 'workspace', 'n_edges', 'status',
                 'want_forces', 'want_virial', '_views')

    def __getitem__(self, name):
        v = self._views.get(name)
        if v is None:
            lay, N, B = self.lay, self.N, self.B
            if name == 'energy':
                v = self.f32[lay.energy:lay.energy + B]
            elif name == 'forces':
                v = self.f32[lay.forces:lay.forces + 3 * N].view(N, 3) if self.want_forces else None
            elif name == 'virial':
                v = self.f32[lay.virial:lay.virial + 9 * B].view(B, 3, 3) if (self.want_forces and self.want_virial) else None
            elif name == 'atom_energy':
                v = self.f32[lay.atom_energy:lay.atom_energy + N]
            elif name == 'atom_node':
                v = self.atom_node
            elif name == 'force_node':
                v = self.force_node
            elif name == 'workspace':
                v = self.workspace
            else:
                raise KeyError(name)
            self._views[name] = v
        return v

    @property
    def row_ptr(self):
        return self.i32[self.lay.row_ptr:self.lay.row_ptr + self.N + 1]

    @property
    def col(self):
        return self.i32[self.lay.col:self.lay.col + self.n_edges]

    @property
    def edge_index(self):
        # RadiusGraph's [2][E] array, made when a caller asks for it (most evaluation steps never do): receiver = the row of the
        # edge, sender = col (nnhip_edge_index_from_csr)
        if self.ei is None:
            E, lay = self.n_edges, self.lay
            self.ei = torch.empty(2 * E, dtype=torch.int64, device=self.i32.device)
            _check(lib().nnhip_edge_index_from_csr(_ptr(self.i32[lay.row_ptr:]), _ptr(self.i32[lay.col:]), self.N, E, _ptr(self.ei),
                                                   None, _stream(self.i32.device)), 'nnhip_edge_index_from_csr')
        E = self.n_edges
        return self.ei[:2 * E].view(2, E)


def forward_dev(model: Model, z, pos, cell, batch, cap: int, prepared: torch.Tensor, tail_host_ptr: int, seq: int,
                want_forces: bool, want_virial: bool, workspace: Optional[torch.Tensor], event_handle: int = 0,
                small_molecules: bool = False) -> DevStep:
    """The whole deferred step in one C call (nnhip_forward_dev): neighbor list into arrays of `cap` edges, the (count, status)
    words stored into the pinned slot `tail_host_ptr` by the last neighbor-list kernel, `seq` behind them; energy / forces pipeline.
    Four allocations per step (two arenas, the two node states); `edge_index` is made on demand (DevStep.edge_index)."""
    L = lib()
    dev = pos.device
    N, B = pos.shape[0], cell.shape[0]
    lay = step_layout(N, B, cap)
    st = DevStep()
    st.N, st.B, st.cap, st.lay, st.n_edges, st.status, st._views = N, B, cap, lay, None, 0, {}
    st.want_forces, st.want_virial = want_forces, want_virial
    st.i32 = torch.empty(lay.i32_count, dtype=torch.int32, device=dev)
    st.f32 = torch.empty(lay.f32_count, dtype=torch.float32, device=dev)
    st.ei = None
    st.atom_node = torch.empty(N, NNHIP_F, dtype=torch.float32, device=dev)
    st.force_node = torch.empty(N, 3, NNHIP_F, dtype=torch.float32, device=dev)
    need = L.nnhip_workspace_bytes(N, cap, B, model.n_layers)
    if workspace is None or workspace.numel() < need or workspace.device != dev:
        workspace = torch.empty(max(need, 256), dtype=torch.uint8, device=dev)
    st.workspace = workspace
    a = StepDev()
    a.z, a.pos, a.cell, a.batch = z.data_ptr(), pos.data_ptr(), cell.data_ptr(), batch.data_ptr()
    a.n_atoms, a.n_mol, a.capacity, a.want_forces, a.want_virial = N, B, cap, int(want_forces), int(want_virial)
    a.i32, a.f32, a.edge_index = st.i32.data_ptr(), st.f32.data_ptr(), None
    a.atom_node, a.force_node = st.atom_node.data_ptr(), st.force_node.data_ptr()
    a.workspace, a.workspace_bytes = workspace.data_ptr(), workspace.numel()
    a.prepared, a.prepared_bytes = prepared.data_ptr(), prepared.numel()
    a.tail_host, a.event, a.seq, a.flags = tail_host_ptr, (event_handle or None), seq, (1 if small_molecules else 0)
    _check(L.nnhip_forward_dev(C.byref(model), C.byref(a), _stream(dev)), 'nnhip_forward_dev')
    return st


def spatial_order(pos: torch.Tensor, z: torch.Tensor, cutoff: float):
    """(perm, inv, z_perm, pos_perm): the atoms of ONE big system in Morton order of cells (nnhip_spatial_order; int32 perm / inv on
    pos.device, no host round trip, deterministic: by cell, then by input index)."""
    L = lib()
    L.nnhip_spatial_order_scratch_bytes.restype = C.c_size_t
    L.nnhip_spatial_order_scratch_bytes.argtypes = [C.c_int32]
    L.nnhip_spatial_order.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_float] + [C.c_void_p] * 6
    pos = _f32c(pos, 'pos')
    N, dev = pos.shape[0], pos.device
    perm = torch.empty(N, dtype=torch.int32, device=dev)
    inv = torch.empty(N, dtype=torch.int32, device=dev)
    z_out, pos_out = torch.empty_like(z), torch.empty_like(pos)
    scratch = torch.empty(L.nnhip_spatial_order_scratch_bytes(N), dtype=torch.uint8, device=dev)
    _check(L.nnhip_spatial_order(_ptr(pos), _ptr(z), N, float(cutoff), _ptr(perm), _ptr(inv), _ptr(z_out), _ptr(pos_out),
                                 _ptr(scratch), _stream(dev)), 'nnhip_spatial_order')
    return perm, inv, z_out, pos_out


def permute_rows(x: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """out[k] = x[idx[k]] over dim 0 (idx int32): the per-atom results of a permuted step in the caller's order."""
    L = lib()
    L.nnhip_permute_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    x = _f32c(x, 'x')
    width = 1
    for d in x.shape[1:]:
        width *= int(d)
    out = torch.empty_like(x)
    _check(L.nnhip_permute_rows(_ptr(x), _ptr(idx), x.shape[0], width, _ptr(out), _stream(x.device)), 'nnhip_permute_rows')
    return out


def edge_index_unpermute(row_ptr, col, perm, inv, n_atoms: int, n_edges: int) -> torch.Tensor:
    """The [2][E] int64 neighbor list of a permuted step in the reference's order for the caller's atom order."""
    L = lib()
    L.nnhip_edge_index_unpermute.argtypes = [C.c_void_p] * 4 + [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    dev = perm.device
    ei = torch.empty(2, n_edges, dtype=torch.int64, device=dev)
    scratch = torch.empty(n_atoms + 1 + n_atoms // 1024 + 2, dtype=torch.int32, device=dev)
    _check(L.nnhip_edge_index_unpermute(_ptr(row_ptr), _ptr(col), _ptr(perm), _ptr(inv), n_atoms, n_edges, _ptr(ei), _ptr(scratch),
                                        _stream(dev)), 'nnhip_edge_index_unpermute')
    return ei


def bf16_mlp_launches() -> int:
    """Edge-MLP launches that took the bf16 compute mode (training under torch.autocast(bfloat16)) since the library was loaded."""
    L = lib()
    L.nnhip_bf16_mlp_launches.restype = C.c_int64
    return int(L.nnhip_bf16_mlp_launches())


def mlp_forms() -> dict:
    """Which forms the fused edge-MLP launches of a large batch take (nnhip_mlp_forms): what bench.py's byte model asks."""
    v = lib().nnhip_mlp_forms()
    return {'split': bool(v & 1), 'regw_bwd': bool(v & 2), 'regw_fwd': bool(v & 4), 'regw_single_bwd': bool(v & 8),
            'regw_single_fwd': bool(v & 16)}


def config() -> dict:
    """Every form choice the library makes in this process (nnhip_config), as a dict."""
    import json
    L = lib()
    L.nnhip_config.argtypes = [C.c_char_p, C.c_size_t]
    L.nnhip_config.restype = C.c_int
    buf = C.create_string_buffer(4096)
    _check(L.nnhip_config(buf, 4096), 'nnhip_config')
    return json.loads(buf.value.decode())


def workspace_layout(N: int, E: int, B: int, n_layers: int) -> WsLayout:
    out = WsLayout()
    _check(lib().nnhip_workspace_layout(N, E, B, n_layers, C.byref(out)), 'nnhip_workspace_layout')
    return out


def alloc_outputs(N: int, B: int, dev, want_forces: bool = True, want_virial: bool = False, want_nodes: bool = True) -> dict:
    """The output tensors of one energy_forces call (they depend on N and B only: NewtonNet.forward allocates them while the
    host waits for the edge count)."""
    out = dict()
    out['energy'] = torch.empty(B, dtype=torch.float32, device=dev)
    out['forces'] = torch.empty(N, 3, dtype=torch.float32, device=dev) if want_forces else None
    out['virial'] = torch.empty(B, 3, 3, dtype=torch.float32, device=dev) if (want_virial and want_forces) else None
    out['atom_energy'] = torch.empty(N, dtype=torch.float32, device=dev)
    out['atom_node'] = torch.empty(N, NNHIP_F, dtype=torch.float32, device=dev) if want_nodes else None
    out['force_node'] = torch.empty(N, 3, NNHIP_F, dtype=torch.float32, device=dev) if want_nodes else None
    return out


def energy_forces(model: Model, z: torch.Tensor, pos: torch.Tensor, cell: torch.Tensor, g: Graph, want_forces: bool = True,
                  want_virial: bool = False, want_nodes: bool = True, workspace: Optional[torch.Tensor] = None,
                  out: Optional[dict] = None, prepared: Optional[torch.Tensor] = None):
    """Run the whole hot path.  Returns dict(energy, forces, virial, atom_energy, atom_node, force_node, workspace).
    `out` = the dict of a previous call with the same sizes: its tensors are reused (static addresses for graph capture)."""
    L = lib()
    dev = z.device
    N, E, B = g.n_atoms, g.n_edges, g.n_mol
    need = L.nnhip_workspace_bytes(N, E, B, model.n_layers)
    if workspace is None or workspace.numel() < need or workspace.device != dev:
        workspace = torch.empty(max(need, 256), dtype=torch.uint8, device=dev)
    if out is None:
        out = alloc_outputs(N, B, dev, want_forces, want_virial, want_nodes)
    out['workspace'] = workspace
    pos, cell = _f32c(pos, 'pos'), _f32c(cell, 'cell')
    pair_ptr = getattr(g, 'pair_ptr', None)
    if pair_ptr is not None and E > 0:
        # the same step through the entry point that also takes the per-row pair counts: the row kernels then split a row into the
        # pairs it owns and the others with two scalar loads instead of a ballot over its cols
        _check(L.nnhip_energy_forces_pp(C.byref(model), _ptr(z), _ptr(pos), _ptr(cell), _ptr(g.mol_ptr), _ptr(g.row_ptr),
                                        g.edge_ptr('col'), g.edge_ptr('rev'), g.edge_ptr('pid'), g.edge_ptr('geo'), g.edge_ptr('xg'),
                                        g.edge_ptr('disp'), N, E, B, _ptr(workspace), workspace.numel(), _ptr(out['energy']),
                                        _ptr(out['forces']), _ptr(out['virial']), _ptr(out['atom_energy']), _ptr(out['atom_node']),
                                        _ptr(out['force_node']), _ptr(prepared), _ptr(pair_ptr),
                                        0 if (g.status & STATUS_BIG_MOLECULE) else 1, _stream(dev)),
               'nnhip_energy_forces_pp')
        return out
    _check(L.nnhip_energy_forces(C.byref(model), _ptr(z), _ptr(pos), _ptr(cell), _ptr(g.mol_ptr), _ptr(g.row_ptr), g.edge_ptr('col'),
                                 g.edge_ptr('rev'), g.edge_ptr('pid'), g.edge_ptr('geo'), g.edge_ptr('xg'), g.edge_ptr('disp'), N, E, B,
                                 _ptr(workspace), workspace.numel(), _ptr(out['energy']), _ptr(out['forces']),
                                 _ptr(out['virial']), _ptr(out['atom_energy']), _ptr(out['atom_node']),
                                 _ptr(out['force_node']), _ptr(prepared), _stream(dev)), 'nnhip_energy_forces')
    return out


PRO_NONE, PRO_SILU = 0, 1
EPI_STORE, EPI_BIAS, EPI_DSILU, EPI_ACC = 0, 1, 2, 3


def linear128(A: torch.Tensor, W: torch.Tensor, out: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None,
              H: Optional[torch.Tensor] = None, prologue: int = PRO_NONE, epilogue: int = EPI_STORE) -> torch.Tensor:
    """C = epilogue(prologue(A) @ W.T) for [M,128] x [128,128] on the fp32 matrix cores (csrc/lin128.hip).
    A / out / H may be row-strided views (last dim contiguous)."""
    assert A.dim() == 2 and A.shape[1] == NNHIP_F and A.stride(1) == 1 and A.dtype == torch.float32
    assert W.shape == (NNHIP_F, NNHIP_F) and W.is_contiguous() and W.dtype == torch.float32
    M = A.shape[0]
    if out is None:
        out = torch.empty(M, NNHIP_F, dtype=torch.float32, device=A.device)
    assert out.shape == A.shape and out.stride(1) == 1
    ldh = H.stride(0) if H is not None else 0
    _check(lib().nnhip_linear128(_ptr(A), A.stride(0) if M > 1 else NNHIP_F, _ptr(W), _ptr(out),
                                 out.stride(0) if M > 1 else NNHIP_F, _ptr(bias), _ptr(H), ldh, M, prologue, epilogue,
                                 _stream(A.device)), 'nnhip_linear128')
    return out


def mlp128(X: torch.Tensor, W1: torch.Tensor, W2: torch.Tensor, H: torch.Tensor, Y: torch.Tensor, mode: int = 0,
           accumulate: bool = False) -> torch.Tensor:
    """Fused Linear->SiLU->Linear (mode 0: writes H, Y) or its adjoint (mode 1: reads H); csrc/mlp128.hip."""
    M = X.shape[0]
    _check(lib().nnhip_mlp128(_ptr(X), X.stride(0), _ptr(W1), _ptr(W2), _ptr(H), H.stride(0), _ptr(Y), Y.stride(0), M,
                              mode, 1 if accumulate else 0, _stream(X.device)), 'nnhip_mlp128')
    return Y


def direct_force(atom_node: torch.Tensor, force_node: torch.Tensor, z: torch.Tensor, head, scale,
                 activation: int = 0) -> torch.Tensor:
    """direct_force head (output.py:115-132): head = nn.Sequential(Linear, act, Linear, act, Linear); scale [119,1] or None;
    activation = ACTIVATION_IDS[name]."""
    N = atom_node.shape[0]
    out = torch.empty(N, 3, dtype=torch.float32, device=atom_node.device)
    scratch = torch.empty(3 * max(N, 1) * NNHIP_F, dtype=torch.float32, device=atom_node.device)
    _check(lib().nnhip_direct_force(_ptr(atom_node), _ptr(force_node), _ptr(z), _ptr(head[0].weight), _ptr(head[0].bias),
                                    _ptr(head[2].weight), _ptr(head[2].bias), _ptr(head[4].weight), _ptr(head[4].bias),
                                    _ptr(scale), int(activation), N, _ptr(scratch), _ptr(out), _stream(atom_node.device)),
           'nnhip_direct_force')
    return out


def segment_sum(x: torch.Tensor, row_ptr: torch.Tensor, n_rows: int) -> torch.Tensor:
    """out[i] = sum of x[row_ptr[i]:row_ptr[i+1]] over dim 0 (deterministic CSR scatter-sum)."""
    x = _f32c(x, 'x')
    width = x[0].numel() if x.shape[0] else int(torch.tensor(x.shape[1:]).prod())
    out = torch.empty((n_rows,) + tuple(x.shape[1:]), dtype=torch.float32, device=x.device)
    _check(lib().nnhip_segment_sum(_ptr(x), _ptr(row_ptr), n_rows, width, _ptr(out), _stream(x.device)),
           'nnhip_segment_sum')
    return out


def gather_rows(x: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """out[e] = x[idx[e]] (idx int32)."""
    x = _f32c(x, 'x')
    width = int(torch.tensor(x.shape[1:]).prod())
    out = torch.empty((idx.numel(),) + tuple(x.shape[1:]), dtype=torch.float32, device=x.device)
    _check(lib().nnhip_gather_rows(_ptr(x), _ptr(idx), idx.numel(), width, _ptr(out), _stream(x.device)),
           'nnhip_gather_rows')
    return out


def timers_enable(on, classes=None):
    """Event timers of the library on / off.  `classes` (names of TIMER_CLASSES): only those classes record events."""
    if on and classes:
        mask = 0
        for name in classes:
            mask |= 1 << (TIMER_CLASSES.index(name) + 1)
        _check(lib().nnhip_timers_enable(mask), 'nnhip_timers_enable')
    else:
        _check(lib().nnhip_timers_enable(1 if on else 0), 'nnhip_timers_enable')


def split_products() -> bool:
    """True when the dense kernels use the split-f16 product form (default; NNHIP_MLP_SPLIT=0 selects fp32 MFMA)."""
    return bool(lib().nnhip_split_products())


def timers_read(reset: bool = True):
    ms = (C.c_double * N_TIMER_CLASSES)()
    cnt = (C.c_int64 * N_TIMER_CLASSES)()
    _check(lib().nnhip_timers_read(ms, cnt, 1 if reset else 0), 'nnhip_timers_read')
    return {name: (ms[k], cnt[k]) for k, name in enumerate(TIMER_CLASSES)}
