"""Output heads of the hot path (mirror of newtonnet/models/output.py, energy + gradient-force subset).

The modules are parameter holders / markers with the reference's names: NewtonNet.forward reads them to
decide what the HIP pipeline must produce.  EnergyOutput's three linears run in csrc/lin128.hip and
csrc/edge.hip:head_out_kernel; the gradient force is the analytic reverse sweep in csrc/pipeline.hip
(replacing torch.autograd.grad at output.py:66-73).
"""
from torch import nn

SUPPORTED = ('energy', 'gradient_force', 'direct_force', 'virial', 'stress')


def get_output_by_string(key, n_features=None, activation=None):
    if key == 'energy':
        return EnergyOutput(n_features, activation)
    if key == 'gradient_force':
        return GradientForceOutput()
    if key == 'virial':
        return VirialOutput()
    if key == 'stress':
        return StressOutput()
    if key == 'direct_force':
        return DirectForceOutput(n_features, activation)
    if key in ('hessian', 'charge', 'bec'):
        raise NotImplementedError(
            f'Output type {key} is outside the MI355X hot path (energy / gradient_force / direct_force / virial / stress); '
            f'see DESIGN.md "out of scope"')
    raise NotImplementedError(f'Output type {key} is not implemented yet')


def get_aggregator_by_string(key):
    if key == 'energy':
        return EnergyAggregator()
    if key in ('gradient_force', 'direct_force', 'hessian', 'virial', 'stress', 'charge', 'bec'):
        return NullAggregator()
    raise NotImplementedError(f'Aggregate type {key} is not implemented yet')


class CustomOutputSet:
    """Attribute bag returned by NewtonNet.forward (output.py:51-54).

    `lazy(name, thunk)` registers an attribute that is materialised on first access: the eval-mode hot path uses it for
    the by-products the reference's autograd leaves in the bag (`pos_grad`, `displacement`, `displacement_grad`), so a
    step that never looks at them launches no kernels for them."""
    def __init__(self, **outputs):
        for key, value in outputs.items():
            setattr(self, key, value)

    def lazy(self, name, thunk):
        self.__dict__.setdefault('_lazy', {})[name] = thunk

    def __getattr__(self, name):          # only reached when normal lookup fails
        lazy = self.__dict__.get('_lazy')
        if lazy is not None and name in lazy:
            value = lazy[name]()         # (a thunk that raises -- the deferred checks of the eval step -- stays registered: the
            del lazy[name]               # same error at every touch, not an AttributeError at the second one)
            setattr(self, name, value)
            return value
        raise AttributeError(name)


class DirectProperty(nn.Module):
    pass


class DerivativeProperty(nn.Module):
    def __init__(self):
        super().__init__()
        self.create_graph = False  # set by NewtonNet.train() / .eval()  (newtonnet.py:106-113)


class EnergyOutput(DirectProperty):
    """Linear -> act -> Linear -> act -> Linear(F, 1)  (output.py:88-96); keys layers.{0,2,4}.{weight,bias}."""
    def __init__(self, n_features, activation):
        super().__init__()
        self.layers = nn.Sequential(
            nn.Linear(n_features, n_features), activation,
            nn.Linear(n_features, n_features), activation,
            nn.Linear(n_features, 1))


class DirectForceOutput(DirectProperty):
    """force = sum_f MLP3(atom_node)[f] * force_node[:, :, f]  (output.py:115-132); keys layers.{0,2,4}.{weight,bias}.
    Runs in nnhip_direct_force (csrc/node128.hip)."""
    def __init__(self, n_features, activation):
        super().__init__()
        self.layers = nn.Sequential(
            nn.Linear(n_features, n_features), activation,
            nn.Linear(n_features, n_features), activation,
            nn.Linear(n_features, n_features))


class GradientForceOutput(DerivativeProperty):
    """force = -dE/dpos (output.py:102-113)."""


class VirialOutput(DerivativeProperty):
    """virial = -dE/d(strain) (output.py:154-165)."""


class StressOutput(DerivativeProperty):
    """stress = (dE/d strain) / det(cell) (output.py:167-180)."""


class EnergyAggregator(nn.Module):
    """Per-molecule sum of atomic energies (output.py:245-247); done by mol_energy_kernel."""


class NullAggregator(nn.Module):
    pass
