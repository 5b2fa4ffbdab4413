import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu via gpurun)')


def pytest_collection_modifyitems(config, items):
    """GPU tests fail loudly (not skip) on a GPU box without the HIP library; on a
    CPU-only box they are deselected by `-m "not gpu"`.  If someone runs the whole
    suite without a marker on a CPU box, skip the gpu ones with a clear reason."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container (run through gpurun)')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)
