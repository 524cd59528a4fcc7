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


def probes_built():
    """True when the loaded library was compiled with -DD3R_PROBES (include/dust3r_hip.h d3r_build_has_probes): the ablation kernels and the probe-only
    D3R_* switches exist. The driver's build (__graft_entry__.build()) is the default one; `D3R_PROBES=1 python -m dust3r_amd.build` makes the other."""
    from dust3r_amd._lib import lib
    return bool(lib.d3r_build_has_probes())


def need_probes(what):
    if not probes_built():
        pytest.skip(f'{what}: compiled in probe builds only (D3R_PROBES=1 python -m dust3r_amd.build)')


def probe_arms(every, default):
    """Parameter lists that name probe-only variants: `every` on a probe build, `default` (the variants the product library holds) otherwise -- decided at
    collection time, so that a default build collects what it can run instead of reporting skips."""
    try:
        return list(every) if probes_built() else list(default)
    except Exception:           # library not built yet: collect the product's arms
        return list(default)
