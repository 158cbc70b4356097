import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'autotune: run with the per-shape conv autotuner enabled')


@pytest.fixture(scope='session')
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')


def rel_err(a, b):
    """max |a-b| / max |b| on CPU doubles."""
    import torch
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.fixture(autouse=True)
def _shipped_kernel_configuration(request):
    """Every -m gpu test runs the SHIPPED kernel configuration: the committed per-shape plan table
    (reconvat_amd/tuned_plans.json -- the same tiles bench.py and the scripts run; shapes outside the table, i.e. the small
    fixtures, use the library default tile).  Tests marked `autotune` exercise the on-line tuner instead; the forced-algo
    operator tests cover every kernel family explicitly."""
    if 'gpu' not in request.keywords:
        yield
        return
    from reconvat_amd import ops
    old = ops.AUTOTUNE
    ops.AUTOTUNE = True if 'autotune' in request.keywords else 'table'
    try:
        yield
    finally:
        ops.AUTOTUNE = old
