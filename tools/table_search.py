"""Coordinate descent over plan tables BY STEP TIME (round 5).  The on-line tuner ranks tiles on isolated launches, and two tuner runs differ in
their near-ties; in the two-stream step those near-ties are worth 0.1-0.2 ms (profiles/r05_tuner_insitu.txt).  Given the shipped table and a few
alternative tables (other tuner runs, older tables), this takes the conv entries of ONE U-Net level (all B = 8 entries of one input height) from
each alternative in turn, measures every such variant interleaved with the current best (tools/insitu_ab.measure: fresh bench.py children), adopts
the best variant if it beats the current one by more than --margin ms, and moves on to the next level; then the weight-gradient partitions as one
more group.

    python tools/table_search.py --out gpurun_out/search/best.json alt1.json alt2.json ...
"""
import argparse
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import insitu_ab  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('alts', nargs='+')
ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'search', 'best.json'))
ap.add_argument('--start', default=os.path.join(ROOT, 'reconvat_amd', 'tuned_plans.json'))
ap.add_argument('--reps', type=int, default=3)
ap.add_argument('--margin', type=float, default=0.04)
ap.add_argument('--skip-conv', action='store_true')
args = ap.parse_args()
os.makedirs(os.path.dirname(args.out), exist_ok=True)
cur = json.load(open(args.start))
alts = [(os.path.splitext(os.path.basename(p))[0], json.load(open(p))) for p in args.alts]


def group_of(key):
    f = key.split(',')
    return int(f[2]) if f[1] == '8' else None          # B = 8 entries by input height (one U-Net level; the O&F stack shares 640)


heights = sorted({group_of(k) for k in cur['conv'] if group_of(k) is not None}, reverse=True)
log = []


def dump(doc, name):
    path = os.path.join(os.path.dirname(args.out), name + '.json')
    with open(path, 'w') as fh:
        json.dump(doc, fh, indent=0, sort_keys=True)
        fh.write('\n')
    return path


def best_of(variants, label):
    global cur
    paths = [('cur', dump(cur, 'cur'), '-')] + [(n, dump(d, 'var_' + n), '-') for n, d in variants]
    times = insitu_ab.measure(paths, reps=args.reps, verbose=False)
    med = {n: statistics.median(t) for n, t in times.items()}
    line = f'{label}: ' + '  '.join(f'{n} {m:.3f}' for n, m in med.items())
    winner = min(med, key=med.get)
    if winner != 'cur' and med[winner] < med['cur'] - args.margin:
        cur = dict(variants)[winner]
        line += f'   -> adopted {winner}'
    else:
        line += '   -> kept'
    print(line, flush=True)
    log.append(line)


for h in ([] if args.skip_conv else heights):
    variants = []
    for name, alt in alts:
        doc = json.loads(json.dumps(cur))
        changed = 0
        for k, v in alt['conv'].items():
            if group_of(k) == h and k in doc['conv'] and doc['conv'][k] != v:
                doc['conv'][k] = v
                changed += 1
        if changed:
            variants.append((name, doc))
    if variants:
        best_of(variants, f'conv H={h}')
# the weight-gradient partitions, level by level (key: taps,B,Hv,Wv,Ca,Cb), then the GEMM split-K factors as one group
for h in sorted({group_of(k) for k in cur['wgrad'] if group_of(k) is not None}, reverse=True):
    variants = []
    for name, alt in alts:
        doc = json.loads(json.dumps(cur))
        changed = 0
        for k, v in alt['wgrad'].items():
            if group_of(k) == h and k in doc['wgrad'] and doc['wgrad'][k] != v:
                doc['wgrad'][k] = v
                changed += 1
        if changed:
            variants.append((name, doc))
    if variants:
        best_of(variants, f'wgrad Hv={h}')
variants = []
for name, alt in alts:
    doc = json.loads(json.dumps(cur))
    if alt.get('gemm') and alt['gemm'] != doc['gemm']:
        doc['gemm'] = {k: alt['gemm'].get(k, v) for k, v in doc['gemm'].items()}
        variants.append((name, doc))
if variants:
    best_of(variants, 'GEMM split-K factors')
cur.setdefault('meta', {})['table_search'] = 'tools/table_search.py: per-level coordinate descent by step time over ' + ', '.join(n for n, _ in alts)
dump(cur, os.path.splitext(os.path.basename(args.out))[0])
with open(os.path.splitext(args.out)[0] + '_log.txt', 'w') as fh:
    fh.write('\n'.join(log) + '\n')
