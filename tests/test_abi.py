"""CPU checks of the drop-in boundary: the shared library loads, exports every symbol that
include/reconvat_hip.h declares, and the ctypes prototypes agree with the header (argument counts).
No compute call is made (there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_decls():
    text = open(os.path.join(ROOT, 'include', 'reconvat_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    decls = {}
    for m in re.finditer(r'\b(int|long|const char\*)\s+(rv_\w+)\s*\(([^;]*?)\)\s*;', text, flags=re.S):
        args = m.group(3).strip()
        n = 0 if args in ('', 'void') else len([a for a in args.split(',') if a.strip()])
        decls[m.group(2)] = n
    return decls


@pytest.fixture(scope='module')
def built():
    from reconvat_amd import build
    return build.build()


def test_header_symbols_exported(built):
    lib = ctypes.CDLL(built)
    decls = header_decls()
    assert len(decls) >= 20
    for name in decls:
        assert hasattr(lib, name), f'{name} declared in include/reconvat_hip.h but not exported'


def test_ctypes_signatures_match_header(built):
    from reconvat_amd import _lib
    decls = header_decls()
    assert set(decls) == set(_lib.SIGNATURES), set(decls) ^ set(_lib.SIGNATURES)
    for name, n in decls.items():
        assert len(_lib.SIGNATURES[name][1]) == n, name
    lib = _lib.load()
    assert lib.rv_abi_version() == 1
    # the library records the sources it was built from; load() refuses a library built from other sources (next test)
    assert lib.rv_source_digest().decode() == _lib.source_digest() and len(_lib.source_digest()) == 16
    assert lib.rv_packed_weight_floats(9, 16, 16) == (9 + 16) * 1 * 1 * 64 * 4      # nine tap fragments + the 16 Winograd fragments
    assert lib.rv_packed_weight_floats(9, 8, 16) == 9 * 1 * 1 * 64 * 2
    assert lib.rv_packed_weight_floats(9, 1, 16) == 9 * 16
    assert lib.rv_reduce_workspace_bytes(4096) == 8


def test_stale_library_is_refused(built, monkeypatch):
    """A prebuilt .so whose baked-in source digest differs from the sources next to it must not load (the library is
    git-ignored and ships prebuilt; nothing else ties it to the sources)."""
    from reconvat_amd import _lib, build
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(build, 'source_digest', lambda: '0123456789abcdef')
    monkeypatch.delenv('RECONVAT_HIP_LIB', raising=False)
    monkeypatch.delenv('RV_SKIP_DIGEST_CHECK', raising=False)
    with pytest.raises(RuntimeError, match='rebuild it'):
        _lib.load()


def test_no_cpu_fallback():
    """The product path must fail loudly on CPU tensors -- never route through PyTorch/the oracle."""
    import torch
    from reconvat_amd import ops, UNet_Onset
    x = torch.zeros(1, 4, 4, 16)
    w = torch.zeros(16, 16, 3, 3)
    with pytest.raises(RuntimeError, match='HIP device only'):
        ops.ConvFn.apply(x, w, None, 'c3', None)
    m = UNet_Onset((2, 2), (2, 2), log=True, reconstruction=False, mode='imagewise', spec='Mel')
    batch = {'audio': torch.zeros(1, 32768), 'onset': torch.zeros(1, 64, 88), 'frame': torch.zeros(1, 64, 88)}
    with pytest.raises(RuntimeError, match='HIP device only'):
        m.run_on_batch(batch, None, False)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, 'reconvat_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f
