"""Activation factory mirroring newtonnet/layers/activations.py:5-30.

The HIP kernels fuse SiLU ('swish' / 'silu', the reference default, scripts/config.yml:34) into the
dense-linear prologue/epilogue; the modules built here are parameter-less markers that keep the
reference's nn.Sequential index layout (Linear, act, Linear -> state_dict keys '.0.' and '.2.').
"""
from torch import nn

HIP_FUSED = ('swish', 'silu')
_OTHERS = {'relu': nn.ReLU, 'elu': nn.ELU, 'leaky_relu': nn.LeakyReLU, 'tanh': nn.Tanh, 'sigmoid': nn.Sigmoid,
           'softplus': nn.Softplus, 'gelu': nn.GELU}


def get_activation_by_string(key):
    if key in HIP_FUSED:
        return nn.SiLU()
    if key in _OTHERS:
        return _OTHERS[key]()
    raise NotImplementedError("The activation function '%s' is unknown." % str(key))
