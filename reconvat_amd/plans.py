"""The SHIPPED kernel configuration: per-shape conv tile choices and weight-gradient partitions.

`tuned_plans.json` (next to this file, committed) is the output of `tools/tune_plans.py` on an MI355X: for every conv launch
shape of the BASELINE workloads the tile the on-line tuner measured fastest (`algo` of rv_conv_fwd) and for every
weight-gradient shape the (waves per workgroup, workgroups) partition (rv_conv_wgrad_set_plan).  It is the DEFAULT
configuration of the library -- `bench.py`, the command-line scripts and every `-m gpu` test run exactly this table -- so the
tiles that produce the headline number are the tiles the parity tests ran, every data-parallel rank uses the same tiles (same
fp32 summation order), and results are reproducible run to run.

    ops.AUTOTUNE = 'table'   (default)    shipped table; a shape that is not in it uses the library default tile
    ops.AUTOTUNE = True      RV_AUTOTUNE=1  time every legal tile on the first eager call of a shape (how the table is made)
    ops.AUTOTUNE = False     RV_AUTOTUNE=0  library default tiles everywhere

A shape keyed at another batch size than the table's (B = 1 and 8) borrows the entry of the same layer geometry at the NEAREST
batch size that is not smaller (B = 4 -> the B = 8 entry: band counts / occupancy are those of the larger launch), else the
largest smaller one; tile legality does not depend on B (checked at first use; an illegal tile falls back to the library default).
A shape whose geometry is not in the table at all (other sequence_length, whole-song evaluation) runs the library default tile;
`RV_TUNE_LOG=1` reports every such miss once.
"""
import hashlib
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
PLAN_FILE = os.environ.get('RV_PLAN_FILE', os.path.join(HERE, 'tuned_plans.json'))

_state = {'loaded': False, 'conv': {}, 'conv_nob': {}, 'wgrad': {}, 'wgrad_nob': {}, 'gemm': {}, 'digest': None, 'meta': {}}
_missed = set()
HITS = {'conv': set(), 'wgrad': set(), 'gemm': set()}         # table keys actually used in this process (tests assert coverage with it)


def default_mode():
    e = os.environ.get('RV_AUTOTUNE')
    if e is None or e == 'table':
        return 'table'
    return e not in ('0', 'off', 'false', 'False', '')


def _load():
    if _state['loaded']:
        return
    _state['loaded'] = True
    if not os.path.exists(PLAN_FILE):
        return
    with open(PLAN_FILE, 'rb') as fh:
        raw = fh.read()
    _state['digest'] = hashlib.sha256(raw).hexdigest()[:16]
    doc = json.loads(raw)
    _state['meta'] = doc.get('meta', {})
    for k, v in doc.get('conv', {}).items():
        key = tuple(int(x) for x in k.split(','))           # (mode, B, H, W, cin, cout, ild, old, stats, bnbwd)
        _state['conv'][key] = int(v)
        _state['conv_nob'].setdefault((key[0],) + key[2:], []).append((key[1], key, int(v)))
    for k, v in doc.get('wgrad', {}).items():
        key = tuple(int(x) for x in k.split(','))           # (taps, B, Hv, Wv, Ca, Cb)
        _state['wgrad'][key] = (int(v[0]), int(v[1]))
        _state['wgrad_nob'].setdefault((key[0],) + key[2:], []).append((key[1], key, (int(v[0]), int(v[1]))))
    for k, v in doc.get('gemm', {}).items():
        _state['gemm'][tuple(int(x) for x in k.split(','))] = int(v)   # (M, N, K, batch, a k-fast, b k-fast, act, accumulate) -> splitk


def digest():
    """sha256[:16] of the shipped table (None: no table) -- bench.py prints it, tests pin it."""
    _load()
    return _state['digest']


def meta():
    _load()
    return dict(_state['meta'])


def conv_entries():
    _load()
    return dict(_state['conv'])


def wgrad_entries():
    _load()
    return dict(_state['wgrad'])


def _nearest_batch(cands, b):
    """Entry of the nearest batch size >= b, else of the largest one below (cands: [(B, key, value)])."""
    above = [c for c in cands if c[0] >= b]
    return min(above, key=lambda c: c[0]) if above else max(cands, key=lambda c: c[0])


def _miss(kind, key):
    if os.environ.get('RV_TUNE_LOG') and (kind, key) not in _missed:
        import sys
        _missed.add((kind, key))
        print(f'[plans] {kind} shape {key} is not in the shipped table: library default tile', file=sys.stderr)


def lookup_conv(key):
    """(algo, exact) for a conv launch key (mode, B, H, W, cin, cout, ild, old, stats, bnbwd) or None."""
    _load()
    key = tuple(int(x) for x in key)
    if key in _state['conv']:
        HITS['conv'].add(key)
        return _state['conv'][key], True
    cands = _state['conv_nob'].get((key[0],) + key[2:])
    if cands:
        _, src, algo = _nearest_batch(cands, key[1])
        HITS['conv'].add(src)
        return algo, False
    _miss('conv', key)
    return None


def lookup_wgrad(key):
    """(nw, wgs) for a weight-gradient key (taps, B, Hv, Wv, Ca, Cb) or None."""
    _load()
    key = tuple(int(x) for x in key)
    if key in _state['wgrad']:
        HITS['wgrad'].add(key)
        return _state['wgrad'][key]
    cands = _state['wgrad_nob'].get((key[0],) + key[2:])
    if cands:
        _, src, plan = _nearest_batch(cands, key[1])
        HITS['wgrad'].add(src)
        return plan
    _miss('wgrad', key)
    return None


def gemm_entries():
    _load()
    return dict(_state['gemm'])


def lookup_gemm(key):
    """split-K factor for a GEMM key (M, N, K, batch, A k-fast, B k-fast, act, accumulate) or None."""
    _load()
    key = tuple(int(x) for x in key)
    v = _state['gemm'].get(key)
    if v is not None:
        HITS['gemm'].add(key)
    return v


def dump(conv, wgrad, meta_, path, gemm=None):
    """Write a table: conv {key tuple: algo}, wgrad {key tuple: (nw, wgs)}, gemm {key tuple: splitk}."""
    doc = {'meta': meta_,
           'gemm': {','.join(str(int(x)) for x in k): int(v) for k, v in sorted((gemm or {}).items())},
           'conv': {','.join(str(int(x)) for x in k): int(v) for k, v in sorted(conv.items())},
           'wgrad': {','.join(str(int(x)) for x in k): [int(v[0]), int(v[1])] for k, v in sorted(wgrad.items())}}
    with open(path, 'w') as fh:
        json.dump(doc, fh, indent=0, sort_keys=True)
        fh.write('\n')
