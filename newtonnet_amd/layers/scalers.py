"""Per-element scale / shift tables of the output heads (mirror of newtonnet/layers/scalers.py:5-69).

State-dict layout is the reference's: `scalers.<k>.scale.weight` / `.shift.weight`, each a [119, 1] embedding indexed by
atomic number (row 0 = padding).  On the HIP path the energy head applies them inside head_out_kernel (csrc/edge.hip) and
the direct-force head inside direct_force_tail_kernel (csrc/node128.hip); the module below only owns the parameters and,
in train mode, enters the hand-written training kernels as a device pointer as well (train_fused.py).
"""
import torch
from torch import nn

N_ELEMENTS = 118 + 1

# which tables each output property owns: (has_scale, has_shift)   -- scalers.py:5-24 of the reference
_TABLES = {
    'energy': (True, True),
    'charge': (True, True),
    'direct_force': (True, False),
    'gradient_force': (False, False),
    'hessian': (False, False),
    'virial': (False, False),
    'stress': (False, False),
    'bec': (False, False),
}


def _table(fill: float):
    # built from a constant tensor: consumes no random numbers, so a seeded construction of the whole model draws the same
    # stream as the reference's (tests/test_host.py: same-seed initialisation)
    return nn.Embedding.from_pretrained(torch.full((N_ELEMENTS, 1), fill), freeze=False, padding_idx=0)


class ScaleShift(nn.Module):
    """`scale` / `shift` follow the reference's constructor convention: None = the table does not exist."""
    def __init__(self, scale=None, shift=None):
        super().__init__()
        self.scale = None if scale is None else _table(1.0)
        self.shift = None if shift is None else _table(0.0)

    def forward(self, output, outputs):          # scalers.py:47-58 (eager use; the HIP kernels fuse this)
        if self.scale is not None:
            output = output * self.scale(outputs.z)
        if self.shift is not None:
            output = output + self.shift(outputs.z)
        return output

    def set_scale(self, values):
        self.scale.weight.data = values.reshape(-1, 1)

    def set_shift(self, values):
        self.shift.weight.data = values.reshape(-1, 1)

    def extra_repr(self):
        return f'scale={self.scale is not None}, shift={self.shift is not None}'


def get_scaler_by_string(key):
    if key not in _TABLES:
        raise NotImplementedError(f'Scaler type {key} is not implemented yet')
    has_scale, has_shift = _TABLES[key]
    return ScaleShift(scale=1.0 if has_scale else None, shift=0.0 if has_shift else None)


def set_scaler_by_string(key, scaler, stats, fit_scale=True, fit_shift=True):
    """Load fitted statistics (data.MolecularStatistics) into a scaler, like newtonnet_train.py:88-90 does."""
    entry = stats.get(key)
    if entry is not None:
        if fit_scale and scaler.scale is not None:
            scaler.set_scale(entry['scale'])
        if fit_shift and scaler.shift is not None:
            scaler.set_shift(entry['shift'])
    return scaler
