"""Data ingest either side of the hot path, without ase / torch_geometric.

Mirrors the parts of newtonnet/data/loader.py that feed the model and its scalers:
  parse_xyz            loader.py:167-194   extxyz frames -> z, pos (wrapped), cell (non-pbc rows zeroed), energy, force
  MolecularStatistics  loader.py:197-230   per-element energy shift (least squares over the composition matrix), a single
                                           residual scale for the elements present, mean force norm per element
The statistics feed set_scaler_by_string (newtonnet_amd/layers/scalers.py), exactly as newtonnet_train.py:88-90 does.
Host-side code (numpy / torch CPU): it is not on the GPU path and has no kernel.
"""
from __future__ import annotations

import re
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

_SYMBOLS = ('X H He Li Be B C N O F Ne Na Mg Al Si P S Cl Ar K Ca Sc Ti V Cr Mn Fe Co Ni Cu Zn Ga Ge As Se Br Kr Rb Sr Y Zr '
            'Nb Mo Tc Ru Rh Pd Ag Cd In Sn Sb Te I Xe Cs Ba La Ce Pr Nd Pm Sm Eu Gd Tb Dy Ho Er Tm Yb Lu Hf Ta W Re Os Ir Pt '
            'Au Hg Tl Pb Bi Po At Rn Fr Ra Ac Th Pa U Np Pu Am Cm Bk Cf Es Fm Md No Lr Rf Db Sg Bh Hs Mt Ds Rg Cn Nh Fl Mc Lv '
            'Ts Og').split()
_Z = {s: i for i, s in enumerate(_SYMBOLS)}
_UNITS = {'Ang': 1.0, 'Angstrom': 1.0, 'Bohr': 0.52917721067, 'eV': 1.0, 'Ha': 27.211386024367243, 'Hartree': 27.211386024367243,
          'kcal/mol': 0.04336410390059322, 'kJ/mol': 0.010364269574711572}     # ase.units values (eV, Angstrom base)


@dataclass
class Frame:
    z: np.ndarray        # [n] int
    pos: np.ndarray      # [n,3]
    cell: np.ndarray     # [3,3], rows of non-periodic directions zeroed (loader.py:177)
    energy: Optional[float]
    force: Optional[np.ndarray]


def _parse_header(line: str) -> Dict[str, str]:
    return {k: (v[1:-1] if v.startswith('"') else v) for k, v in re.findall(r'(\w+)=("[^"]*"|\S+)', line)}


def read_extxyz(path: str, max_frames: Optional[int] = None, length_unit: str = 'Ang', energy_unit: str = 'eV') -> List[Frame]:
    """Frames of an (ext)xyz file with Properties=species:S:1:pos:R:3[:forces:R:3], energy=..., Lattice=..., pbc=...
    Positions are wrapped into the cell along periodic directions like atoms.get_positions(wrap=True) (loader.py:174)."""
    lu, eu = _UNITS[length_unit], _UNITS[energy_unit]
    frames: List[Frame] = []
    with open(path) as fh:
        while True:
            line = fh.readline()
            if not line.strip():
                break
            n = int(line)
            hdr = _parse_header(fh.readline())
            props = hdr.get('Properties', 'species:S:1:pos:R:3').split(':')
            cols, c = {}, 0
            for name, _, width in zip(props[0::3], props[1::3], props[2::3]):
                cols[name] = (c, c + int(width))
                c += int(width)
            rows = [fh.readline().split() for _ in range(n)]
            z = np.array([_Z[r[cols['species'][0]]] for r in rows])
            pos = np.array([[float(v) for v in r[cols['pos'][0]:cols['pos'][1]]] for r in rows])
            fkey = 'forces' if 'forces' in cols else ('force' if 'force' in cols else None)
            force = np.array([[float(v) for v in r[cols[fkey][0]:cols[fkey][1]]] for r in rows]) if fkey else None
            cell = np.array([float(v) for v in hdr['Lattice'].split()]).reshape(3, 3) if 'Lattice' in hdr else np.zeros((3, 3))
            pbc = np.array([t in ('T', 'True', 'true', '1') for t in hdr.get('pbc', 'F F F').split()])
            if 'Lattice' in hdr and 'pbc' not in hdr:
                pbc[:] = True
            if pbc.any():
                frac = np.linalg.solve(cell.T, pos.T).T
                frac[:, pbc] %= 1.0
                pos = frac @ cell
            cell = cell.copy()
            cell[~pbc] = 0.0
            energy = float(hdr['energy']) * eu if 'energy' in hdr else None
            frames.append(Frame(z=z, pos=pos * lu, cell=cell * lu, energy=energy,
                                force=None if force is None else force * eu / lu))
            if max_frames and len(frames) >= max_frames:
                break
    return frames


def collate(frames: Sequence[Frame], dtype=torch.float32, device='cpu'):
    """(z, pos, cell, batch, energy, force) laid out like a PyG Batch (ase_interface.py:131-142, trainer.py:305-306)."""
    z = torch.tensor(np.concatenate([f.z for f in frames]), dtype=torch.long, device=device)
    pos = torch.tensor(np.concatenate([f.pos for f in frames]), dtype=dtype, device=device)
    cell = torch.tensor(np.stack([f.cell for f in frames]), dtype=dtype, device=device)
    batch = torch.tensor(np.concatenate([np.full(len(f.z), b) for b, f in enumerate(frames)]), dtype=torch.long, device=device)
    energy = (torch.tensor([f.energy for f in frames], dtype=dtype, device=device)
              if all(f.energy is not None for f in frames) else None)
    force = (torch.tensor(np.concatenate([f.force for f in frames]), dtype=dtype, device=device)
             if all(f.force is not None for f in frames) else None)
    return z, pos, cell, batch, energy, force


class MolecularStatistics:
    """Callable mirror of loader.py:197-230.  stats['energy'] = {'shift': [119], 'scale': [119]}, stats['force'] = {'scale'}."""
    def __call__(self, z, batch, energy=None, force=None):
        stats = {}
        z = z.long().cpu()
        batch = batch.long().cpu()
        z_unique = z.unique()
        if energy is not None:
            energy = energy.cpu()
            n_mol = int(batch.max()) + 1
            onehot = torch.nn.functional.one_hot(z, int(z.max()) + 1).to(energy.dtype)
            formula = torch.zeros(n_mol, onehot.shape[1], dtype=energy.dtype).index_add_(0, batch, onehot)
            solution = torch.linalg.lstsq(formula, energy, driver='gelsd').solution
            shift = torch.zeros(119, dtype=energy.dtype)
            shift[z_unique] = solution[z_unique]
            std = ((energy - formula @ solution).square().sum() / formula.sum()).sqrt()
            scale = torch.ones(119, dtype=energy.dtype)
            scale[z_unique] = std
            stats['energy'] = {'shift': shift, 'scale': scale}
        if force is not None:
            fn = force.cpu().norm(dim=-1)
            sums = torch.zeros(int(z.max()) + 1, dtype=fn.dtype).index_add_(0, z, fn)
            cnt = torch.zeros(int(z.max()) + 1, dtype=fn.dtype).index_add_(0, z, torch.ones_like(fn)).clamp(min=1)
            fscale = torch.ones(119, dtype=fn.dtype)
            fscale[z_unique] = (sums / cnt)[z_unique]
            stats['force'] = {'scale': fscale}
        return stats
