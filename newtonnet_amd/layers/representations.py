"""Parameter holders for the edge embedding (newtonnet/layers/representations.py:5-43, 206-238).

The arithmetic (radius graph, scaled norm, p=9 polynomial envelope, Bessel basis) lives in
csrc/graph.hip; these modules only carry the hyper-parameters and the frozen `frequencies`
parameter under the reference's state_dict name
`embedding_layers.edge_embedding.embedding.frequencies`.
"""
import math

import torch
from torch import nn


class RadialBesselLayer(nn.Module):
    def __init__(self, n_basis):
        super().__init__()
        self.n_basis = n_basis
        self.frequencies = nn.Parameter(torch.arange(1, n_basis + 1) * math.pi, requires_grad=False)

    def __repr__(self):
        return f'{self.__class__.__name__}(basis={self.n_basis})'


class EdgeEmbedding(nn.Module):
    def __init__(self, cutoff, n_basis=20):
        super().__init__()
        self.cutoff = float(cutoff)
        self.n_basis = n_basis
        self.envelope_p = 9          # PolynomialCutoff(p=9), representations.py:17
        self.embedding = RadialBesselLayer(n_basis=n_basis)

    def __repr__(self):
        return f'{self.__class__.__name__}(r={self.cutoff}, p={self.envelope_p}, basis={self.n_basis}) [HIP]'
