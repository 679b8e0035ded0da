import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'lab: GEMM tilings that exist in the tuning build only (python -m drvae_amd.build '
                            '--lab, DRVAE_HIP_LIB=build_lab/libdrvae_lab.so); run with -m "gpu and lab"')


def pytest_collection_modifyitems(config, items):
    """the lab tilings are not in the product library: their cases are deselected (not skipped) unless asked for, so
    that a skip in the GPU record is always a real one"""
    if 'lab' in (config.getoption('-m') or '') or os.environ.get('DRVAE_HIP_LIB'):
        return
    drop = [it for it in items if it.get_closest_marker('lab') is not None]
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = [it for it in items if it.get_closest_marker('lab') is None]


@pytest.fixture(scope='session')
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU visible')
    return torch.device('cuda:0')
