import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _drain_gpu_between_tests():
    """runners march one batch ahead on a side stream: finish all device work before the next test re-uses the memory"""
    yield
    import gc
    import torch
    if torch.cuda.is_available():
        from jnerf_amd.utils.config import get_cfg
        get_cfg().clear()
        gc.collect()
        torch.cuda.synchronize()
