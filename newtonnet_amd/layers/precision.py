"""String -> dtype map (newtonnet/layers/precision.py:3-13), plus bfloat16."""
import torch


def get_precision_by_string(key):
    if key in ['float32', 'float', 'single']:
        return torch.float32
    if key in ['float64', 'double']:
        return torch.float64
    if key in ['float16', 'half']:
        return torch.float16
    if key in ['bfloat16', 'bf16']:
        return torch.bfloat16
    raise ValueError(f'precision {key} is not supported')
