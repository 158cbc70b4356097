"""HBM traffic of the conv phase of ONE training step from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in
separate runs of `bench.py --no-graph`, CSV output).  Step boundaries are the adam_k dispatches; the last complete step
is summed over the conv kernel family.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half
of the bytes of wide coalesced reads -> doubled; WRITE_SIZE is taken as reported (uncalibrated).

    python tools/pmc_traffic.py <dir with *_counter_collection.csv> > profiles/r01_pmc_traffic.json
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

def _src_digest():
    """Digest of the kernel sources the profiled library was built from (reconvat_amd/build.py::source_digest)."""
    import os as _os
    sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
    from reconvat_amd import build as _b
    return _b.source_digest()


CONV = ('conv3x3_lds_k', 'conv3x3_wino_k', 'conv3x3_wino2_k', 'conv_mfma_k', 'conv_small_k', 'conv_narrow_out_k', 'conv_cin12_k', 'wgrad_mfma_k', 'wgrad_wino_k', 'wgrad_small_k', 'wgrad_small_sw_k', 'wgrad_cin1_k', 'wgrad_reduce_k',
        'wgrad_reduce_table_k')


# round 5: every family of the step, same names as tools/rocpd_stats.py (bench.py prints measured / algorithmic bytes per family)
FAMILIES = [
    ('conv3x3 fwd+dgrad (MFMA)', ('conv3x3_lds_k', 'conv3x3_wino_k', 'conv3x3_wino2_k')),
    ('conv direct 3x3/1x1/2x2 (MFMA / HBM)', ('conv_mfma_k',)),
    ('conv C<=2 layers (HBM)', ('conv_small_k', 'wgrad_small_k', 'conv_narrow_out_k', 'conv_cin12_k', 'wgrad_small_sw_k', 'wgrad_cin1_k')),
    ('weight gradients (MFMA)', ('wgrad_mfma_k', 'wgrad_wino_k', 'wgrad_reduce', 'sums_fold')),
    ('linear GEMMs (MFMA)', ('gemm_', 'colsum', 'sigmoid', 'zero_strided')),
    ('local attention', ('attn_',)),
    ('BatchNorm + leaky-ReLU (HBM)', ('bn_',)),
    ('front-end (HBM)', ('mel_',)),
    ('losses / VAT elementwise (HBM)', ('reduce_', 'loss_bwd', 'vat_', 'elementwise', 'vectorized', 'Philox', 'distribution')),
    ('optimiser + weight repack', ('adam_k', 'pack_', 'clip', 'counter_add')),
]


def family_pass(path, counter):
    """{family: (reported KB, launches)} over the last complete optimiser step of one PMC pass."""
    per = defaultdict(float)
    name = {}
    for row in csv.DictReader(open(path)):
        if row['Counter_Name'] != counter:
            continue
        d = int(row['Dispatch_Id'])
        per[d] += float(row['Counter_Value'])
        name[d] = row['Kernel_Name']
    ids = sorted(per)
    adam = [d for d in ids if name[d].startswith('adam_k')]
    lo, hi = adam[-2], adam[-1]
    fam = defaultdict(lambda: [0.0, 0])
    for d in ids:
        if lo < d <= hi:
            key = next((f for f, pats in FAMILIES if any(p in name[d] for p in pats)), 'other')
            fam[key][0] += per[d]
            fam[key][1] += 1
    return fam


def one_pass(path, counter):
    per = defaultdict(float)
    name = {}
    for row in csv.DictReader(open(path)):
        if row['Counter_Name'] != counter:
            continue
        d = int(row['Dispatch_Id'])
        per[d] += float(row['Counter_Value'])
        name[d] = row['Kernel_Name']
    ids = sorted(per)
    adam = [d for d in ids if name[d].startswith('adam_k')]
    if len(adam) < 2:
        raise SystemExit(f'{path}: fewer than two optimiser steps recorded')
    lo, hi = adam[-2], adam[-1]
    tot, n, fam = 0.0, 0, defaultdict(float)
    for d in ids:
        if lo < d <= hi and any(k in name[d] for k in CONV):
            tot += per[d]
            n += 1
            fam[next(k for k in CONV if k in name[d])] += per[d]
    return tot, n, fam


def main():
    root = sys.argv[1]
    out, fams = {}, {}
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        for f in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
            with open(f) as fh:
                head = fh.read(200000)
            if counter in head:
                kb, n, fam = one_pass(f, counter)
                out[counter] = {'kb_reported': kb, 'launches': n, 'by_kernel_kb': dict(fam)}
                fams[counter] = family_pass(f, counter)
                break
    families = {}
    for fam in sorted(set(fams['FETCH_SIZE']) | set(fams['WRITE_SIZE'])):
        fk, fn = fams['FETCH_SIZE'].get(fam, (0.0, 0))
        wk, _ = fams['WRITE_SIZE'].get(fam, (0.0, 0))
        families[fam] = {'fetch_bytes': fk * 1024 * 2, 'write_bytes': wk * 1024, 'traffic_bytes': fk * 1024 * 2 + wk * 1024, 'launches': fn}
    fetch = out['FETCH_SIZE']['kb_reported'] * 1024 * 2       # gfx950: x2
    write = out['WRITE_SIZE']['kb_reported'] * 1024
    import subprocess
    try:
        git = subprocess.check_output(['git', 'rev-parse', '--short', 'HEAD'], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:  # noqa: BLE001 -- the GPU box has no .git
        git = os.environ.get('RV_GIT_SHA', 'unknown (no .git on the GPU box)')
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from reconvat_amd import plans
    res = {'scope': 'all conv-family launches of one optimiser step (eager launches, B_l=B_ul=8)', 'git': git,
           'kernel_plan_table': plans.digest(), 'source_digest': _src_digest(),         # the tile table these launches ran (bench.py flags a mismatch with its own)
           'fetch_bytes': fetch, 'write_bytes': write, 'traffic_bytes': fetch + write,
           'launches': out['FETCH_SIZE']['launches'],
           'families': families,          # every family of the step (eager single-stream launches), gfx950-corrected like the conv total
           'correction': 'FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported',
           'raw': out}
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
