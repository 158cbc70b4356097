"""Tolerances of the VAT-carrying loss terms, derived from the REFERENCE's own arithmetic noise.

tests/golden/lds_spread.npz (make_golden.py g_lds_spread) holds, for every VAT-carrying fixture, the reference's loss values at
8 threads fp32 (the golden), 1 thread fp32 and fp64.  At XI = 1e-6 the power-iteration direction is rounding-noise driven, so
these terms move by 3e-4 .. 5e-3 between those runs of the SAME reference code (non-VAT terms: < 1e-6).  A HIP loss term is
accepted when |hip - reference| <= max(1e-3, 2 x spread) x |reference| (north_star's 1e-3, widened only where the reference
itself is measurably noisier), and every measured error is appended to gpurun_out/parity_errors.jsonl for DESIGN.md section 4.
"""
import json
import os

import numpy as np

_G = os.path.join(os.path.dirname(__file__), 'golden', 'lds_spread.npz')
_OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'parity_errors.jsonl')
_cache = {}


def _gold():
    if 'g' not in _cache:
        _cache['g'] = np.load(_G)
    return _cache['g']


def is_vat_key(key):
    return 'LDS' in key or 'r_norm' in key


def spread(case, key=None):
    """Relative spread of the VAT terms of `case` (e.g. 'onset_T64'): the WORST of the case's VAT keys.  All of them are
    functions of the same rounding-noise-driven adversarial direction, and each per-key figure is only a two-sample
    estimate (1 thread, fp64) of that noise -- e.g. r_norm_ul moved by 1.2e-3 in case onset_T64_step and by 5.3e-3 in the
    sibling case onset_T64 -- so the case-level maximum is the stable measure of "the reference's own noise"."""
    g = _gold()
    keys = [str(k) for k in g[case + '_keys']]
    sp = g[case + '_spread']
    return float(max(s for k, s in zip(keys, sp) if is_vat_key(k)))


def tol(case, key):
    if not is_vat_key(key):
        return 1e-3
    # round 3: 2 x (was 3 x) -- every one of the 68 measured (case, key) errors of the final build is <= 1.75 x the case's spread
    return max(1e-3, 2.0 * spread(case, key))


def check(case, key, got, ref, where):
    """Assert one loss term and log the measured error."""
    err = abs(float(got) - float(ref)) / max(abs(float(ref)), 1e-6)
    t = tol(case, key)
    try:
        os.makedirs(os.path.dirname(_OUT), exist_ok=True)
        with open(_OUT, 'a') as fh:
            fh.write(json.dumps({'where': where, 'case': case, 'key': key, 'hip': float(got), 'reference': float(ref),
                                 'rel_err': err, 'tol': t, 'ref_spread': spread(case, key) if is_vat_key(key) else None}) + '\n')
    except OSError:
        pass
    assert err <= t, (where, case, key, float(got), float(ref), err, t)
    return err
