"""Precision names accepted by the trainer / calculator configs -> torch dtypes.

Same vocabulary and error as the reference's helper (newtonnet/layers/precision.py:3-13); bfloat16 added for the bf16
autocast training path.  The HIP inference path itself computes in float32 only.
"""
import torch

_DTYPES = {
    torch.float32: ('float32', 'float', 'single'),
    torch.float64: ('float64', 'double'),
    torch.float16: ('float16', 'half'),
    torch.bfloat16: ('bfloat16', 'bf16'),
}
_BY_NAME = {name: dtype for dtype, names in _DTYPES.items() for name in names}


def get_precision_by_string(key):
    try:
        return _BY_NAME[key]
    except (KeyError, TypeError):
        raise ValueError(f'precision {key} is not supported') from None
