"""ASE calculator for the MI355X hot path -- mirror of newtonnet/utils/ase_interface.py:18-142.

Same constructor, `implemented_properties`, `calculate()` contract and result shapes as the reference's
MLAseCalculator.  `ase` is optional at import time: when it is installed the class derives from
ase.calculators.calculator.Calculator; otherwise a minimal stand-in base keeps the same attributes
(`results`, `atoms`) so that any object exposing get_atomic_numbers / get_positions(wrap=True) / get_cell /
get_pbc works (this image has no ase; the parity tests drive it with such an object against K1).
"""
from __future__ import annotations

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


def _is_single(atoms) -> bool:
    return hasattr(atoms, 'get_positions')


class MLAseCalculator(_Base):
    implemented_properties = ['energy', 'free_energy', 'forces', 'stress']
    # the reference additionally lists 'charges', 'bec', 'hessian' (ase_interface.py:19): outside the hot path

    def __init__(self, model_path, properties: list = None, device: str = None, precision: str = 'float32',
                 **kwargs):
        """
        model_path: path of a whole-module pickle (torch.save(model), trainer.py:219) of a newtonnet_amd NewtonNet,
                    a state_dict file (.pt/.npz with the reference's key names), or a NewtonNet instance.
        properties: subset of implemented_properties; default: what the model predicts.
        device:     'cuda' (default when available).  The HIP path has no CPU implementation.
        precision:  'float32' / 'single' (the HIP path computes in fp32).
        """
        _Base.__init__(self, **kwargs)
        self.device = torch.device(device) if device is not None else torch.device(
            'cuda' if torch.cuda.is_available() else 'cpu')
        self.dtype = get_precision_by_string(precision)
        self.properties = properties
        self.model = self.load_model(model_path)

    # ------------------------------------------------------------------ ase_interface.py:52-81
    def calculate(self, atoms=None, properties=None, system_changes=None):
        _Base.calculate(self, atoms, self.properties, system_changes)
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

    # ------------------------------------------------------------------ ase_interface.py:83-129
    def load_model(self, model):
        from newtonnet_amd.models import NewtonNet
        if isinstance(model, torch.nn.Module):
            pass
        elif str(model).endswith('.npz'):
            with np.load(model) as f:
                sd = {k: torch.from_numpy(f[k]) for k in f.files}
            model = self._from_state_dict(sd)
        else:
            obj = torch.load(model, map_location='cpu', weights_only=False)
            model = self._from_state_dict(obj) if isinstance(obj, dict) else obj
        if not isinstance(model, NewtonNet):
            raise TypeError(f'expected a newtonnet_amd NewtonNet, got {type(model)}')
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
    def _from_state_dict(sd):
        from newtonnet_amd.models import NewtonNet
        n_layers = 0
        while f'interaction_layers.{n_layers}.equiv_update.weight' in sd:
            n_layers += 1
        F = sd['embedding_layers.node_embedding.weight'].shape[1]
        nb = sd['embedding_layers.edge_embedding.embedding.frequencies'].numel()
        model = NewtonNet(n_features=F, n_basis=nb, n_interactions=n_layers,
                          output_properties=['energy', 'gradient_force'])
        model.to(torch.float64)
        model.load_state_dict({k: v.to(torch.float64) for k, v in sd.items()})
        return model

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
