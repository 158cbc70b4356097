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
def _deterministic_conv_choice(request):
    """The per-shape conv autotuner picks by wall time, so its choice (and with it the fp32 summation order) can differ
    from run to run.  Parity tests use the library's default kernel choice; tests marked `autotune` (and the forced-algo
    operator tests) cover the tuner and every kernel family explicitly."""
    if 'gpu' not in request.keywords:
        yield
        return
    from reconvat_amd import ops
    old = ops.AUTOTUNE
    ops.AUTOTUNE = 'autotune' in request.keywords
    try:
        yield
    finally:
        ops.AUTOTUNE = old
