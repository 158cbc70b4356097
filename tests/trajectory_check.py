"""Checker of a K-step training trajectory against tests/golden/trajectory.npz (the REFERENCE's own train_VAT_model run,
model/helper_functions.py:570-615; generator g_trajectory of tests/golden/make_golden.py).  Shared by the CPU oracle test and the
GPU test of the product's TrainStep + FlatAdam.

The trajectory amplifies rounding noise (Adam's first updates are lr * sign(g); the reference's own fp32 run drifts from its fp64 run
by 1-2 % of a parameter tensor and by 1e-2 of a loss term within six steps), so every quantity is held to the reference's OWN drift:

    losses of iteration i :  |x - f64| / |f64| <= 2 x (worst fp32-vs-fp64 movement of that iteration's terms) + 1e-3
    parameters, Adam moments, BatchNorm running statistics (relative L2 on the stored samples, against the fp64 run):
                             e <= 2 x e_ref32 + 1e-3        (e_ref32 = the reference's fp32 run against its fp64 run; for tensors of fewer
                                                             than 64 values at least the median e_ref32 of the tensors of that kind)
    learning rate of every iteration and after the last step, num_batches_tracked: exact.
"""
import json
import os

import numpy as np

G = os.path.join(os.path.dirname(__file__), 'golden', 'trajectory.npz')
_cache = {}


def gold():
    if 'g' not in _cache:
        _cache['g'] = np.load(G)
    return _cache['g']


def digest(t, n):
    f = t.detach().double().flatten().cpu()
    stride = max(1, f.numel() // n)
    return np.concatenate([[f.norm().item()], f[::stride][:n].numpy()])


def check(tag, losses, lrs, params, adam_m, adam_v, buffers, n, where, log=None):
    """losses: [K][nkeys] floats in the reference's key order; lrs: the K rates used + the rate after step K; params / adam_m / adam_v /
    buffers: name -> tensor after step K (adam_* only for parameters that received a gradient).  Returns the report rows."""
    g = gold()
    rows, bad = [], []
    want_lr = g[tag + '_lr']
    assert len(lrs) == len(want_lr) and np.allclose(np.asarray(lrs, dtype=np.float64), want_lr, rtol=1e-6, atol=0), (list(lrs), list(want_lr))
    l32, l64 = g[tag + '_f32_losses'], g[tag + '_f64_losses']
    assert np.asarray(losses).shape == l64.shape, (np.asarray(losses).shape, l64.shape)
    keys = [str(k) for k in g[tag + '_keys']]
    for i in range(l64.shape[0]):
        spread = float(np.max(np.abs(l32[i] - l64[i]) / np.maximum(np.abs(l64[i]), 1e-6)))
        for j, k in enumerate(keys):
            e = abs(float(losses[i][j]) - l64[i][j]) / max(abs(l64[i][j]), 1e-6)
            rows.append({'what': 'loss', 'iteration': i, 'key': k, 'e': e, 'bar': 2 * spread + 1e-3, 'ref_spread': spread})
            if e > 2 * spread + 1e-3:
                bad.append(rows[-1])
    nograd = set(str(k) for k in g[tag + '_nograd'])
    for kind, have, nn in (('p', params, n), ('m', adam_m, n // 2), ('v', adam_v, n // 4)):
        names = [k[len(f'{tag}_f64_{kind}:'):] for k in g.files if k.startswith(f'{tag}_f64_{kind}:')]
        assert names and set(names) <= set(have), (kind, sorted(set(names) - set(have))[:5])
        # tensors whose every value is rounding noise (e.g. the moments of a conv bias that feeds a train-mode BatchNorm: the true gradient is
        # zero) have no meaningful relative error: the floor of the denominator is 1e-3 of the LARGEST rms among the tensors of this kind
        rms = {k: g[f'{tag}_f64_{kind}:' + k][0] / max(np.sqrt(np.prod(have[k].shape)), 1.0) for k in names}
        floor_rms = 1e-3 * max(rms.values())
        def ref_err(k):
            d64, d32 = g[f'{tag}_f64_{kind}:' + k], g[f'{tag}_f32_{kind}:' + k].astype(np.float64)
            return np.linalg.norm(d32[1:] - d64[1:]) / max(np.linalg.norm(d64[1:]), floor_rms * np.sqrt(len(d64) - 1))
        pooled = float(np.median([ref_err(k) for k in names]))
        for k in names:
            d64, d32 = g[f'{tag}_f64_{kind}:' + k], g[f'{tag}_f32_{kind}:' + k].astype(np.float64)
            d = digest(have[k], nn)
            assert len(d) == len(d64), (kind, k, len(d), len(d64))
            den = max(np.linalg.norm(d64[1:]), floor_rms * np.sqrt(len(d64) - 1))
            e, e_ref = np.linalg.norm(d[1:] - d64[1:]) / den, np.linalg.norm(d32[1:] - d64[1:]) / den
            if len(d64) - 1 < 64:
                # a tensor of a few values (biases of the 1- and 2-channel heads): its own fp32-vs-fp64 figure is ONE draw of the noise,
                # not an estimate of it -- the yardstick is at least the median drift of the tensors of this kind
                e_ref = max(e_ref, pooled)
            nden = max(d64[0], floor_rms * np.sqrt(np.prod(have[k].shape)))
            en, en_ref = abs(d[0] - d64[0]) / nden, abs(d32[0] - d64[0]) / nden
            row = {'what': kind, 'key': k, 'e': e, 'e_ref32': e_ref, 'bar': 2 * e_ref + 1e-3, 'e_norm': en, 'e_norm_ref32': en_ref}
            rows.append(row)
            if k in nograd and kind == 'p':
                assert e == 0.0 or e <= 1e-7, ('a never-touched parameter moved', k, e)
            if e > 2 * e_ref + 1e-3 or en > 2 * max(en_ref, e_ref) + 1e-3:
                bad.append(row)
    for k in g.files:
        if not k.startswith(f'{tag}_f64_s:'):
            continue
        name = k[len(f'{tag}_f64_s:'):]
        b64, b32 = g[k], g[f'{tag}_f32_s:' + name]
        got = buffers[name].detach().double().cpu().numpy()
        if name.endswith('num_batches_tracked'):
            assert int(got) == int(b64) == int(b32), (name, int(got), int(b64))
            continue
        den = max(np.linalg.norm(b64), 1e-30)
        e, e_ref = np.linalg.norm(got - b64) / den, np.linalg.norm(b32 - b64) / den
        row = {'what': 'bn', 'key': name, 'e': e, 'e_ref32': e_ref, 'bar': 2 * e_ref + 1e-3}
        rows.append(row)
        if e > 2 * e_ref + 1e-3:
            bad.append(row)
    if log:
        try:
            os.makedirs(os.path.dirname(log), exist_ok=True)
            with open(log, 'w') as fh:
                json.dump({'where': where, 'tag': tag, 'rows': rows}, fh)
        except OSError:
            pass
    assert not bad, (where, tag, bad[:6])
    return rows


def summary(rows):
    out = {}
    for what in ('loss', 'p', 'm', 'v', 'bn'):
        r = [x for x in rows if x['what'] == what]
        if r:
            out[what] = {'n': len(r), 'worst_share_of_bar': max(x['e'] / x['bar'] for x in r), 'median_e': float(np.median([x['e'] for x in r])),
                         'median_e_ref32': float(np.median([x.get('e_ref32', x.get('ref_spread', 0.0)) for x in r]))}
    return out
