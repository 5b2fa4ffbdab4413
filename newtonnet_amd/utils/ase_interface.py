"""ASE calculator for the MI355X hot path -- mirror of newtonnet/utils/ase_interface.py:18-142.

Same constructor, `implemented_properties`, `calculate()` contract and result shapes as the reference's
MLAseCalculator.  `ase` is optional at import time: when it is installed the class derives from
ase.calculators.calculator.Calculator; otherwise a minimal stand-in base keeps the same attributes
(`results`, `atoms`) so that any object exposing get_atomic_numbers / get_positions(wrap=True) / get_cell /
get_pbc works (this image has no ase; the parity tests drive it with such an object against K1).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from newtonnet_amd.layers.precision import get_precision_by_string
from newtonnet_amd.layers.scalers import get_scaler_by_string
from newtonnet_amd.models.output import (DerivativeProperty, get_aggregator_by_string, get_output_by_string)

try:  # pragma: no cover - ase is not in the build image
    from ase.calculators.calculator import Calculator as _Base
    _HAVE_ASE = True
except Exception:  # noqa: BLE001
    _HAVE_ASE = False

    class _Base:
        """Just enough of ase.calculators.calculator.Calculator for calculate()."""
        def __init__(self, **kwargs):
            self.results = {}
            self.atoms = None
            self.parameters = dict(kwargs)

        def calculate(self, atoms=None, properties=None, system_changes=None):
            self.atoms = atoms


def _current_key(k: str) -> str:
    """Parameter names of the pre-2.0 checkpoint layout (scripts/md17_model/training_1/models/best_model.pt) -> current."""
    k = k.replace('embedding_layer.edge_embedding.frequencies', 'embedding_layers.edge_embedding.embedding.frequencies')
    return k.replace('embedding_layer.', 'embedding_layers.')


class _ReferenceUnpickler(__import__('pickle').Unpickler):
    """Whole-module pickles written by the reference name classes of the `newtonnet` package (and of `les`).  When those
    import, they are used as they are (and converted through state_dict()); when they do not -- this package does not depend
    on the reference -- each missing class becomes an attribute-only nn.Module stand-in: nn.Module's own __setstate__
    restores `_parameters` / `_modules`, which is all state_dict() needs."""
    def find_class(self, module, name):
        try:
            return super().find_class(module, name)
        except (ImportError, AttributeError):
            if module.split('.')[0] in ('newtonnet', 'les'):
                return type(name, (torch.nn.Module,), {'__module__': module})
            raise


class _ReferencePickle:
    """pickle_module for torch.load (needs Unpickler and load)."""
    __name__ = 'newtonnet_amd_reference_pickle'
    Unpickler = _ReferenceUnpickler

    @staticmethod
    def load(f, **kw):
        return _ReferenceUnpickler(f, **kw).load()


def _is_single(atoms) -> bool:
    return hasattr(atoms, 'get_positions')


class MLAseCalculator(_Base):
    implemented_properties = ['energy', 'free_energy', 'forces', 'stress']
    # the reference additionally lists 'charges', 'bec', 'hessian' (ase_interface.py:19): outside the hot path

    def __init__(self, model_path, properties: list = None, device: str = None, precision: str = 'float32',
                 skin: float = 0.5, capture: bool = False, **kwargs):
        """
        model_path: path of a whole-module pickle (torch.save(model), trainer.py:219) of a newtonnet_amd NewtonNet,
                    a state_dict file (.pt/.npz with the reference's key names), or a NewtonNet instance.
        properties: subset of implemented_properties; default: what the model predicts.
        device:     'cuda' (default when available).  The HIP path has no CPU implementation.
        precision:  'float32' / 'single' (the HIP path computes in fp32).
        skin:       Verlet skin in Angstrom for the single-structure (MD-loop) path: the neighbor list is built once with
                    cutoff + skin and reused until an atom has moved more than skin / 2; candidates outside the cutoff at a
                    given step contribute exactly zero, so the results are the exact-list results up to fp32 summation order.
                    0 rebuilds the exact list on every call (the reference's behaviour, ase_interface.py:52-81).
        capture:    replay the step of the MD-loop path from a HIP graph (one launch instead of ~45).  Off by default:
                    on ROCm 7.2 / MI355X the replay measured the same 0.43 ms per step as direct launches (the ~45
                    dependent dispatches cost ~8 us each on the GPU either way) and every list rebuild re-captures.
        """
        _Base.__init__(self, **kwargs)
        self.device = torch.device(device) if device is not None else torch.device(
            'cuda' if torch.cuda.is_available() else 'cpu')
        self.dtype = get_precision_by_string(precision)
        self.properties = properties
        self.skin = float(skin)
        self.capture = bool(capture)
        self._md = None          # state of the MD-loop path (list, static buffers, captured graph)
        self.md_stats = {'steps': 0, 'rebuilds': 0}
        self.model = self.load_model(model_path)

    # ------------------------------------------------------------------ ase_interface.py:52-81
    def calculate(self, atoms=None, properties=None, system_changes=None):
        _Base.calculate(self, atoms, self.properties, system_changes)
        if _is_single(atoms) and self.skin > 0 and self.device.type == 'cuda':
            return self._calculate_md(atoms)
        if _is_single(atoms):
            atoms = [atoms]
        z, pos, cell, batch = self.format_data(atoms)
        n_frames, n_atoms = len(atoms), len(atoms[0])
        pred = self.model(z, pos, cell, batch)
        if 'energy' in self.properties:
            self.results['energy'] = pred.energy.cpu().detach().numpy().squeeze()
        if 'free_energy' in self.properties:
            self.results['free_energy'] = pred.energy.cpu().detach().numpy().squeeze()
        if 'forces' in self.properties:
            force = pred.gradient_force.cpu().detach().numpy()
            self.results['forces'] = force.reshape(n_frames, n_atoms, 3).squeeze()
        if 'stress' in self.properties:
            stress = pred.stress.cpu().detach().numpy()
            self.results['stress'] = stress[:, [0, 1, 2, 1, 0, 0], [0, 1, 2, 2, 2, 1]].squeeze()
        del pred

    # ------------------------------------------------------------------ MD-loop latency path
    def _calculate_md(self, atoms):
        """One structure per call, called thousands of times by an MD driver (simulate.py:21-30): keep everything that does
        not change between steps on the device -- the candidate neighbor list (cutoff + skin), the parameter struct, the
        workspace and output buffers -- re-evaluate only the edge geometry, and replay the launches from a HIP graph."""
        from newtonnet_amd import hip
        z = np.asarray(atoms.get_atomic_numbers())
        pos = np.asarray(atoms.get_positions(wrap=True), dtype=np.float64)
        try:
            free = np.asarray(atoms.get_positions(), dtype=np.float64)       # unwrapped: what the skin criterion follows
        except TypeError:
            free = pos
        cell = np.array(getattr(atoms.get_cell(), 'array', atoms.get_cell()), dtype=np.float64).reshape(3, 3).copy()
        cell[~np.asarray(atoms.get_pbc(), dtype=bool)] = 0.0
        st = self._md
        stale = st is None or st['z'].shape != z.shape or not np.array_equal(st['z'], z) \
            or not np.array_equal(st['cell'] != 0, cell != 0)
        if not stale:
            # Verlet criterion with a moving cell (NPT): a pair inside the cutoff now was inside cutoff + skin at build time
            # as long as  2 max|dr| + strain * (cutoff + skin) <= skin   (dr: unwrapped displacement since the build)
            moved = float(np.sqrt(np.max(np.sum((free - st['ref']) ** 2, axis=1))))
            strain = 0.0
            if np.any(cell != 0) and not np.array_equal(st['cell'], cell):
                ref = np.where(st['cell'].any(axis=1, keepdims=True), st['cell'], np.eye(3))
                cur = np.where(cell.any(axis=1, keepdims=True), cell, np.eye(3))
                strain = float(np.linalg.norm(np.linalg.solve(ref, cur) - np.eye(3), 2))
            stale = 2.0 * moved + strain * (st['cutoff'] + self.skin) > self.skin
        if stale:
            st = self._md_build(z, pos, free, cell)
        elif not np.array_equal(st['cell_now'], cell):
            st['cell_now'] = cell.copy()
            st['cell_dev'].copy_(torch.tensor(cell[None], dtype=torch.float32), non_blocking=False)
        # (a big system is kept in the spatial order of its last list build: _md_build)
        order = st['order']
        st['pos_host'].copy_(torch.from_numpy((pos if order is None else pos[order]).astype(np.float32)))
        if not st['zero_copy_in']:
            st['pos'].copy_(st['pos_host'], non_blocking=True)
        if st['graph'] is not None:
            st['graph'].replay()
        else:
            self._md_step(st)
        if not st['zero_copy']:
            st['out_host'].copy_(st['buf'], non_blocking=True)
        torch.cuda.current_stream().synchronize()
        self.md_stats['steps'] += 1
        n = len(z)
        res = st['out_host'].numpy()
        if 'energy' in self.properties:
            self.results['energy'] = res[0].copy()
        if 'free_energy' in self.properties:
            self.results['free_energy'] = res[0].copy()
        if 'forces' in self.properties:
            if st['order'] is None:
                self.results['forces'] = res[1:1 + 3 * n].reshape(n, 3).copy()
            else:
                forces = np.empty((n, 3), dtype=res.dtype)
                forces[st['order']] = res[1:1 + 3 * n].reshape(n, 3)
                self.results['forces'] = forces
        if 'stress' in self.properties:
            stress = -res[1 + 3 * n:10 + 3 * n].reshape(3, 3) / np.float32(np.linalg.det(cell))
            self.results['stress'] = stress[[0, 1, 2, 1, 0, 0], [0, 1, 2, 2, 2, 1]]

    def _md_build(self, z, pos, free, cell):
        from newtonnet_amd import hip
        from newtonnet_amd.models.output import StressOutput, VirialOutput
        dev, model = self.device, self.model
        emb = model.embedding_layers.edge_embedding
        n = len(z)
        want_forces = any(isinstance(l, DerivativeProperty) for l in model.output_layers)
        want_virial = any(isinstance(l, (VirialOutput, StressOutput)) for l in model.output_layers)
        st = dict(z=z.copy(), cell=cell.copy(), cell_now=cell.copy(), ref=free.copy(), graph=None, want_forces=want_forces,
                  want_virial=want_virial, order=None)
        # one big system: the list, the device arrays and every step until the next rebuild use the atoms in Morton order of
        # cells (hip.spatial_order -- partner rows close together whatever order the Atoms object
        # has); positions are gathered and forces scattered on the host, energy and stress are sums
        from newtonnet_amd.models import newtonnet as nn_mod
        order_min = model.__dict__.get('_spatial_order_min', nn_mod._SPATIAL_ORDER_MIN)
        if order_min > 0 and n >= order_min:
            # (hip.spatial_order: the library's own kernels, csrc/graph.hip)
            st['order'] = hip.spatial_order(torch.tensor(pos, dtype=torch.float32, device=dev),
                                            torch.tensor(z, dtype=torch.long, device=dev), float(emb.cutoff))[0].cpu().numpy().astype(np.int64)
            z, pos = z[st['order']], pos[st['order']]
        st['z_dev'] = torch.tensor(z, dtype=torch.long, device=dev)
        st['pos'] = torch.tensor(pos, dtype=torch.float32, device=dev)
        st['cell_dev'] = torch.tensor(cell[None], dtype=torch.float32, device=dev)
        st['batch'] = torch.zeros(n, dtype=torch.long, device=dev)
        st['pos_host'] = torch.tensor(pos, dtype=torch.float32).pin_memory()
        # (reading the positions in place from the pinned host array as well measured no gain: 393 vs 388 us; off by default)
        st['zero_copy_in'] = os.environ.get('NNHIP_MD_ZERO_COPY_IN', '0') != '0'
        pos_dev = st['pos']
        if st['zero_copy_in']:
            st['pos'] = st['pos_host']
        st['freq'] = emb.embedding.frequencies
        st['cutoff'] = float(emb.cutoff)
        st['model'] = model._hip_model(list(model.output_properties).index('energy'))
        st['g'] = hip.build_graph(pos_dev, st['cell_dev'], st['batch'], st['cutoff'] + self.skin, st['freq'], cell_host=cell[None],
                                  envelope=emb.envelope_id)
        st['prep'] = hip.prepare(st['model'], dev)     # parameters are fixed while the calculator owns the model (eval)
        g = st['g']
        # Results (energy, forces, virial: a few hundred bytes) are written by the last kernels STRAIGHT into pinned host memory
        # (hipHostMalloc memory is mapped into the device's address space on ROCm): no device->host copy per step, whose DMA
        # start-up latency was ~80 us of the ~390 us step.  NNHIP_MD_ZERO_COPY=0 restores the device buffer + copy.
        st['zero_copy'] = os.environ.get('NNHIP_MD_ZERO_COPY', '1') != '0'
        st['out_host'] = torch.zeros(1 + 3 * n + 9, dtype=torch.float32).pin_memory()
        st['buf'] = st['out_host'] if st['zero_copy'] else torch.zeros(1 + 3 * n + 9, dtype=torch.float32, device=dev)
        st['out'] = dict(energy=st['buf'][0:1], forces=st['buf'][1:1 + 3 * n].view(n, 3) if want_forces else None,
                         virial=st['buf'][1 + 3 * n:].view(1, 3, 3) if want_virial else None,
                         atom_energy=torch.empty(n, dtype=torch.float32, device=dev), atom_node=None, force_node=None)
        need = hip.lib().nnhip_workspace_bytes(n, g.n_edges, 1, st['model'].n_layers)
        st['ws'] = torch.empty(max(need, 256), dtype=torch.uint8, device=dev)
        self._md_step(st)                      # warm-up on the current stream (one-time kernel attributes, lazy loads)
        if self.capture:
            try:
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    self._md_step(st)
                torch.cuda.current_stream().wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, capture_error_mode='thread_local'):
                    self._md_step(st)
                st['graph'] = graph
            except Exception as exc:  # noqa: BLE001 -- capture is an optimisation; fall back to plain launches
                import warnings
                warnings.warn(f'HIP graph capture of the MD step failed ({exc}); using direct launches')
                st['graph'] = None
                torch.cuda.synchronize()
        self._md = st
        self.md_stats['rebuilds'] += 1
        return st

    def _md_step(self, st):
        from newtonnet_amd import hip
        hip.refresh_graph(st['g'], st['pos'], st['cell_dev'], st['batch'], st['cutoff'], st['freq'])
        hip.energy_forces(st['model'], st['z_dev'], st['pos'], st['cell_dev'], st['g'], want_forces=st['want_forces'],
                          want_virial=st['want_virial'], want_nodes=False, workspace=st['ws'], out=st['out'],
                          prepared=st['prep'])

    # ------------------------------------------------------------------ ase_interface.py:83-129
    def load_model(self, model):
        """model: a NewtonNet of this package, ANY nn.Module with the reference's parameter names (e.g. the reference's own
        NewtonNet), a state_dict, or the path of a whole-module pickle (torch.save(model), trainer.py:219 -- written by
        this package OR by the reference, ase_interface.py:87), a state_dict file or an .npz of arrays.  Anything that is
        not already a newtonnet_amd NewtonNet is rebuilt from its state_dict (the key names are identical by construction)."""
        from newtonnet_amd.models import NewtonNet
        if isinstance(model, torch.nn.Module):
            pass
        elif isinstance(model, dict):
            model = self._from_state_dict(model)
        elif str(model).endswith('.npz'):
            with np.load(model) as f:
                sd = {k: torch.from_numpy(f[k]) for k in f.files}
            model = self._from_state_dict(sd)
        else:
            obj = torch.load(model, map_location='cpu', weights_only=False, pickle_module=_ReferencePickle)
            model = self._from_state_dict(obj) if isinstance(obj, dict) else obj
        if not isinstance(model, NewtonNet):
            if not isinstance(model, torch.nn.Module):
                raise TypeError(f'expected a NewtonNet module, a state_dict or a path, got {type(model)}')
            model = self._from_module(model)
        if self.properties is None:
            self.properties = [{'energy': 'energy', 'gradient_force': 'forces', 'stress': 'stress'}[k]
                               for k in model.output_properties if k in ('energy', 'gradient_force', 'stress')]
        else:
            key_map = {'energy': 'energy', 'free_energy': 'energy', 'forces': 'gradient_force', 'stress': 'stress'}
            keys_to_keep = ['energy']
            for prop in self.properties:
                if prop not in key_map:
                    raise NotImplementedError(f"property '{prop}' is outside the MI355X hot path")
                key = key_map[prop]
                keys_to_keep.append(key)
                if key in model.output_properties:
                    continue
                model.output_properties.append(key)
                model.output_layers.append(get_output_by_string(key))
                model.scalers.append(get_scaler_by_string(key))
                model.aggregators.append(get_aggregator_by_string(key))
            for i in reversed([i for i, k in enumerate(model.output_properties) if k not in keys_to_keep]):
                model.output_properties.pop(i)
                model.output_layers.pop(i)
                model.scalers.pop(i)
                model.aggregators.pop(i)
        model.to(self.dtype)
        model.to(self.device)
        model.eval()
        model.embedding_layers.requires_dr = any(isinstance(l, DerivativeProperty) for l in model.output_layers)
        return model

    @staticmethod
    def _from_state_dict(sd, cutoff: float = 5.0, activation: str = 'swish', output_properties=None):
        """Rebuild a NewtonNet from the reference's parameter names; the sizes are read off the tensors.  Keys of the
        pre-2.0 layout of the shipped MD17 checkpoint (`embedding_layer.*`) are renamed."""
        from newtonnet_amd.models import NewtonNet
        sd = {_current_key(k): v for k, v in sd.items()}
        n_layers = 0
        while f'interaction_layers.{n_layers}.equiv_update.weight' in sd:
            n_layers += 1
        F = sd['embedding_layers.node_embedding.weight'].shape[1]
        nb = sd['embedding_layers.edge_embedding.embedding.frequencies'].numel()
        props = list(output_properties) if output_properties else ['energy', 'gradient_force']
        model = NewtonNet(cutoff=cutoff, n_features=F, n_basis=nb, n_interactions=n_layers, activation=activation,
                          layer_norm='interaction_layers.0.layer_norm.weight' in sd, output_properties=props)
        model.to(torch.float64)
        model.load_state_dict({k: v.to(torch.float64) for k, v in sd.items()})
        return model

    @classmethod
    def _from_module(cls, obj):
        """A module that is not this package's NewtonNet -- the reference's own class, or the attribute-only stand-in the
        unpickler builds when the reference package is not importable: read the hyper-parameters the state_dict does not
        carry (cutoff, activation, output heads) off the module tree, then rebuild from state_dict()."""
        sd = {k: v.detach().cpu() for k, v in obj.state_dict().items()}
        emb = getattr(obj, 'embedding_layers', None) or getattr(obj, 'embedding_layer', None)
        cutoff = 5.0
        ee = getattr(emb, 'edge_embedding', None)
        for holder, attr in ((getattr(ee, 'norm', None), 'r'), (getattr(ee, 'radius_graph', None), 'r'), (ee, 'cutoff')):
            if holder is not None and hasattr(holder, attr):
                cutoff = float(getattr(holder, attr))
                break
        activation = 'swish'
        layers = getattr(obj, 'interaction_layers', None)
        if layers is not None and len(layers):
            act = layers[0].message_nodepart[1]
            names = {'SiLU': 'swish', 'ReLU': 'relu', 'ELU': 'elu', 'LeakyReLU': 'leaky_relu', 'Tanh': 'tanh',
                     'Sigmoid': 'sigmoid', 'Softplus': 'softplus', 'GELU': 'gelu', 'ShiftedSoftplus': 'ssp'}
            if type(act).__name__ not in names:
                raise NotImplementedError(f'activation module {type(act).__name__} is outside the MI355X hot path')
            activation = names[type(act).__name__]
        props = [k for k in getattr(obj, 'output_properties', ['energy', 'gradient_force'])]
        for k in props:
            if k not in ('energy', 'gradient_force', 'direct_force', 'virial', 'stress'):
                raise NotImplementedError(f"output property '{k}' of the loaded model is outside the MI355X hot path")
        return cls._from_state_dict(sd, cutoff=cutoff, activation=activation, output_properties=props)

    # ------------------------------------------------------------------ ase_interface.py:131-142
    def format_data(self, atoms_list):
        zs, ps, cs, bs = [], [], [], []
        for b, atoms in enumerate(atoms_list):
            z = np.asarray(atoms.get_atomic_numbers())
            pos = np.asarray(atoms.get_positions(wrap=True), dtype=np.float64)
            cell = np.array(getattr(atoms.get_cell(), 'array', atoms.get_cell()), dtype=np.float64).reshape(3, 3).copy()
            pbc = np.asarray(atoms.get_pbc(), dtype=bool)
            cell[~pbc] = 0.0
            zs.append(z)
            ps.append(pos)
            cs.append(cell)
            bs.append(np.full(len(z), b))
        z = torch.tensor(np.concatenate(zs), dtype=torch.long, device=self.device)
        pos = torch.tensor(np.concatenate(ps), dtype=self.dtype, device=self.device)
        cell = torch.tensor(np.stack(cs), dtype=self.dtype, device=self.device)
        batch = torch.tensor(np.concatenate(bs), dtype=torch.long, device=self.device)
        return z, pos, cell, batch
