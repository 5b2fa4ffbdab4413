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


class PolynomialCutoff(nn.Module):
    """Parameter holder of the polynomial envelope (representations.py:138-171); evaluated in csrc/graph.hip:envelope_eval."""
    def __init__(self, p):
        super().__init__()
        self.p = int(p)

    def __repr__(self):
        return f'{self.__class__.__name__}(p={self.p})'


class CosineCutoff(nn.Module):
    """Behler cosine cutoff y = (1 + cos(pi x)) / 2 (representations.py:177-203).  The reference's EdgeEmbedding always builds
    PolynomialCutoff(9) (:17); assigning `edge_embedding.envelope = CosineCutoff()` -- on the reference or here -- switches the
    radial-filter tables and the edge embedding to it."""
    def __repr__(self):
        return f'{self.__class__.__name__}()'


class EdgeEmbedding(nn.Module):
    def __init__(self, cutoff, n_basis=20):
        super().__init__()
        self.cutoff = float(cutoff)
        self.n_basis = n_basis
        self.envelope = PolynomialCutoff(p=9)          # representations.py:17
        self.embedding = RadialBesselLayer(n_basis=n_basis)

    @property
    def envelope_id(self) -> int:
        """envelope selector of the C ABI: p > 0 for PolynomialCutoff(p), NNHIP_ENVELOPE_COSINE (-1) for CosineCutoff"""
        env = self.__dict__.get('_modules', {}).get('envelope', None)
        if env is None or type(env).__name__ == 'PolynomialCutoff':
            p = int(getattr(env, 'p', 9))
            if not 1 <= p <= 64:
                raise NotImplementedError(f'PolynomialCutoff(p={p}) is outside the HIP kernels (1..64)')
            return p
        if type(env).__name__ == 'CosineCutoff':
            return -1
        raise NotImplementedError(f'envelope {type(env).__name__} is outside the MI355X hot path')

    def __repr__(self):
        return f'{self.__class__.__name__}(r={self.cutoff}, envelope={self.envelope}, basis={self.n_basis}) [HIP]'
