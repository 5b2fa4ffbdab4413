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


def triclinic_fuzz_inputs(n, seed=2024, cutoff=5.0):
    """n two-atom periodic 'molecules', each in its own random general triclinic cell (lattice vectors 8..15 A long, every
    component drawn at random, |det| >= 200 A^3), placed where the reference's neighbor predicate is decided by roundings
    (representations.py:85-98: solve -> round -> bmm -> norm < r, all in fp32):
      kind 0: fractional separation +-0.5 +- k ulp (k in -4..4) along one lattice vector, random perpendicular offset;
      kind 1: minimum-image distance r (1 +- k ulp) at a random fractional separation;
      kind 2: both at once -- +-0.5 +- k ulp along a lattice vector shorter than 2 r, with the perpendicular offset that puts
              the pair at distance r (1 +- k' ulp).
    Pure numpy (PCG64) in float64, rounded once to float32: the same arrays in gen_golden.py (reference side) and in the GPU
    test.  Returns pos [2n,3] f32, cell [n,3,3] f32, batch [2n] i64, kind [n] i8."""
    rng = np.random.Generator(np.random.PCG64(seed))
    pos = np.zeros((2 * n, 3), dtype=np.float32)
    cells = np.zeros((n, 3, 3), dtype=np.float32)
    kinds = np.zeros(n, dtype=np.int8)
    r = float(cutoff)

    def ulps(x, k):
        v = np.float32(x)
        for _ in range(abs(int(k))):
            v = np.nextafter(v, np.float32(np.inf if k > 0 else -np.inf))
        return np.float64(v)
    for b in range(n):
        while True:
            c = rng.normal(size=(3, 3))
            c *= (rng.uniform(8.0, 15.0, size=3) / np.linalg.norm(c, axis=1))[:, None]
            if abs(np.linalg.det(c)) >= 200.0:
                break
        c32 = c.astype(np.float32)
        c = c32.astype(np.float64)
        kind = int(rng.integers(0, 3))
        axis = int(rng.integers(0, 3))
        a = c[axis]
        if kind == 2 and np.linalg.norm(a) >= 2.0 * r - 0.05:
            kind = 0
        p0 = rng.uniform(0.0, 1.0, size=3) @ c
        k1, k2 = int(rng.integers(-4, 5)), int(rng.integers(-4, 5))
        sgn = 1.0 if rng.random() < 0.5 else -1.0
        perp = np.cross(a, rng.normal(size=3))
        perp /= np.linalg.norm(perp)
        if kind == 0:
            d = sgn * ulps(0.5, k1) * a + perp * rng.uniform(0.0, 6.0)
        elif kind == 1:
            frac = rng.uniform(-0.45, 0.45, size=3)
            v = frac @ c
            d = v / np.linalg.norm(v) * ulps(r, k2)
        else:
            half = sgn * ulps(0.5, k1) * a
            d = half + perp * np.sqrt(max(ulps(r, k2) ** 2 - half @ half, 0.0))
        p032 = p0.astype(np.float32)
        pos[2 * b] = p032
        pos[2 * b + 1] = (p032.astype(np.float64) + d).astype(np.float32)
        cells[b] = c32
        kinds[b] = kind
    batch = np.repeat(np.arange(n, dtype=np.int64), 2)
    return pos, cells, batch, kinds


def direct_train_state(dtype=torch.float64):
    """state_dict of the ['energy', 'direct_force'] model behind tests/golden/case_train_direct.npz: the seeded shared parameters
    + the direct_force head / scaler the reference's constructor drew (stored in the fixture)."""
    c = load_npz('case_train_direct.npz')
    sd = load_state('rand', dtype)
    for k, v in c.items():
        if k.startswith('state.'):
            sd[k[6:]] = torch.from_numpy(v).to(dtype)
    return sd, c
