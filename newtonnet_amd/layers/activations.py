"""Activation factory: the names of newtonnet/layers/activations.py:5-30 and what runs them here.

Every name but 'swiglu' is fused into the HIP kernels (csrc/common.h: act_f / dact_f, ids in include/newtonnet_hip.h);
'swish' / 'silu' -- the default of every published config (scripts/config.yml:34) -- keep a dedicated fast path.  The modules
built here keep the reference's nn.Sequential index layout (Linear, act, Linear -> state_dict keys '.0.' and '.2.') and are
the reference's own forward for anyone who calls it.  ('swiglu' cannot be constructed by the reference's own factory either: its
class needs constructor arguments the factory does not pass.)
"""
import math

import torch
from torch import nn

HIP_FUSED = ('swish', 'silu', 'relu', 'elu', 'leaky_relu', 'tanh', 'sigmoid', 'softplus', 'gelu', 'ssp')


class ShiftedSoftplus(nn.Module):
    """softplus(x) - ln 2 (activations.py:33-46)."""
    def forward(self, x):
        return torch.nn.functional.softplus(x) - math.log(2.0)


_MODULES = {'swish': nn.SiLU, 'silu': nn.SiLU, 'relu': nn.ReLU, 'elu': nn.ELU, 'leaky_relu': nn.LeakyReLU, 'tanh': nn.Tanh,
            'sigmoid': nn.Sigmoid, 'softplus': nn.Softplus, 'gelu': nn.GELU, 'ssp': ShiftedSoftplus}


def get_activation_by_string(key):
    if key in _MODULES:
        return _MODULES[key]()
    raise NotImplementedError("The activation function '%s' is unknown." % str(key))
