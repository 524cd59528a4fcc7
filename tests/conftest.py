import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs an AMD gfx950 GPU (run by the driver with -m gpu on an MI355X)')


@pytest.fixture(scope='session')
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU visible')
    from dust3r_amd import _lib
    _lib.require_device()
    return torch.device('cuda:0')


@pytest.fixture(scope='session', autouse=True)
def _cpu_threads():
    """The CPU oracle runs inside the GPU tests too: size torch's thread pool by measurement, not by os.cpu_count()
    (256 logical CPUs on the GPU boxes, of which the container can use far fewer)."""
    from oracle import tune_threads
    n = tune_threads()
    print(f'[conftest] torch CPU threads = {n}')
    yield
