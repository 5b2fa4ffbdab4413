"""Per-element scale/shift (newtonnet/layers/scalers.py:5-69).  Applied inside head_out_kernel (csrc/edge.hip)."""
import torch
from torch import nn


def get_scaler_by_string(key):
    table = {'energy': (1.0, 0.0), 'gradient_force': (None, None), 'direct_force': (1.0, None),
             'hessian': (None, None), 'virial': (None, None), 'stress': (None, None), 'charge': (0.1, 0.0),
             'bec': (None, None)}
    if key not in table:
        raise NotImplementedError(f'Scaler type {key} is not implemented yet')
    scale, shift = table[key]
    return ScaleShift(scale=scale, shift=shift)


def set_scaler_by_string(key, scaler, stats, fit_scale=True, fit_shift=True):
    if scaler.scale is not None and key in stats and fit_scale:
        scaler.set_scale(stats[key]['scale'])
    if scaler.shift is not None and key in stats and fit_shift:
        scaler.set_shift(stats[key]['shift'])
    return scaler


class ScaleShift(nn.Module):
    def __init__(self, scale=None, shift=None):
        super().__init__()
        self.scale = (nn.Embedding.from_pretrained(torch.ones(118 + 1, 1), freeze=False, padding_idx=0)
                      if scale is not None else None)
        self.shift = (nn.Embedding.from_pretrained(torch.zeros(118 + 1, 1), freeze=False, padding_idx=0)
                      if shift is not None else None)

    def set_scale(self, scale):
        self.scale.weight.data = scale.reshape(-1, 1)

    def set_shift(self, shift):
        self.shift.weight.data = shift.reshape(-1, 1)

    def __repr__(self):
        return f'{self.__class__.__name__}(scale={self.scale is not None}, shift={self.shift is not None})'
