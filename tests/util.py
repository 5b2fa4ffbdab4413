"""Shared helpers for the test-suite: fixture loading (tests/golden) and tolerances."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

# Stated fp32 tolerances vs the fp64 oracle (SURVEY.md section 7 "fp32 noise floor")
FORCE_MAE_TOL = 1e-5      # eV/A
FORCE_MAX_TOL = 5e-5      # eV/A
ENERGY_ULPS = 2           # fp32 ulps of |E|


def load_npz(name):
    with np.load(os.path.join(GOLDEN, name)) as f:
        return {k: f[k] for k in f.files}


def load_state(which, dtype=torch.float64):
    """'rand' -> seeded reference-initialised weights, 'ckpt' -> the shipped MD17-aspirin model."""
    fn = {'rand': 'rand_state_seed0.npz', 'ckpt': 'ckpt_state.npz'}[which]
    return {k: torch.from_numpy(v).to(dtype) for k, v in load_npz(fn).items()}


def case_inputs(case, dtype=torch.float64):
    c = load_npz(f'case_{case}.npz')
    return (torch.from_numpy(c['z']).long(), torch.from_numpy(c['pos']).to(dtype),
            torch.from_numpy(c['cell']).to(dtype), torch.from_numpy(c['batch']).long(), c)


def energy_tol(e_ref):
    """2 fp32 ulps of |E| (at least 1e-5 eV for small energies)."""
    e = np.abs(np.asarray(e_ref, dtype=np.float64))
    return np.maximum(ENERGY_ULPS * np.spacing(e.astype(np.float32)).astype(np.float64), 1e-5)
