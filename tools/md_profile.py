"""Kernel trace of the calculator's MD-loop path (one aspirin molecule): run under rocprofv3 --kernel-trace."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from newtonnet_amd.models import NewtonNet
from newtonnet_amd.utils import MLAseCalculator
from tests.test_ase_calculator import FakeAtoms
torch.manual_seed(0)
model = NewtonNet(output_properties=['energy', 'gradient_force']).to('cuda'); model.eval()
z, pos, cell, batch = bench.synthetic_aspirin(1, 0, 'cpu')
numbers, p0 = z.numpy(), pos.double().numpy()
vel = np.random.default_rng(0).normal(0, 0.002, p0.shape)
calc = MLAseCalculator(model, properties=['energy', 'forces'], device='cuda', skin=0.5)
for s in range(60): calc.calculate(FakeAtoms(numbers, p0 + s * vel))
print(calc.md_stats)
