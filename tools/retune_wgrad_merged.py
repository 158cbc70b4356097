"""Partitions for the MERGED weight-gradient launches (ops.WgradMerger: the four backward passes of a transcriber layer in one rv_conv_wgrad_seg
launch, i.e. the B = 8 geometry at 32 images): for every B = 8 weight-gradient entry of a plan table, time the candidate partitions at
--images images (the on-line tuner of ops._tune_wgrad on one contiguous tensor -- the kernels do not care where a segment lives) and add the
winner as a new entry.  Without such an entry a merged launch borrows the B = 8 partition.

    python tools/retune_wgrad_merged.py --out gpurun_out/plans_wgrad_merged.json [--images 32]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from reconvat_amd import ops, _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--table', default=os.path.join(ROOT, 'reconvat_amd', 'tuned_plans.json'))
ap.add_argument('--out', required=True)
ap.add_argument('--images', type=int, default=32)
args = ap.parse_args()
doc = json.load(open(args.table))
dev = torch.device('cuda:0')
lib = _lib.load()
ops.AUTOTUNE = True
nb = args.images
added = 0
for key, old in sorted(doc['wgrad'].items()):
    taps, bb, hv, wv, ca, cb = (int(v) for v in key.split(','))
    if bb != 8 or ca <= 2 or cb <= 2:
        continue
    # a layer with this reduction geometry: taps 9 -> 3x3 (U = x), 1 -> 1x1, 4 -> 2x2 stride 2 (U = the fine tensor: x of the down conv / dY of the up conv)
    kind = {9: 'c3', 1: 'c1', 4: 'down'}[taps]
    hu, wu = (hv, wv) if taps != 4 else (2 * hv, 2 * wv)
    x = torch.randn(nb, hu, wu, ca, device=dev)
    dy = torch.randn(nb, hv, wv, cb, device=dev)
    w = torch.randn(cb, ca, *{9: (3, 3), 1: (1, 1), 4: (2, 2)}[taps], device=dev)
    picks = {}
    for rnd in range(2):
        ops._wgrad_tuned.discard((taps, nb, hv, ca, cb))
        ops.conv_wgrad(kind, x, dy, w, True)
        torch.cuda.synchronize()
        plan = ops._wgrad_plans.get((taps, nb, hv, wv, ca, cb))
        us = ops._tune_us.get(('wgrad', (taps, nb, hv, wv, ca, cb)))
        picks[tuple(plan)] = min(picks.get(tuple(plan), 1e9), us)
    plan = min(picks, key=picks.get)
    nk = f'{taps},{nb},{hv},{wv},{ca},{cb}'
    print(f'{nk}: {list(plan)}  {picks[plan]:.1f} us   (B = 8 entry: {old})', flush=True)
    doc['wgrad'][nk] = [int(plan[0]), int(plan[1])]
    added += 1
    del x, dy
    torch.cuda.empty_cache()
doc.setdefault('meta', {})['wgrad_merged'] = f'tools/retune_wgrad_merged.py: partitions of the merged weight-gradient launches at {nb} images'
with open(args.out, 'w') as fh:
    json.dump(doc, fh, indent=0, sort_keys=True)
    fh.write('\n')
print(f'{added} entries added -> {args.out}')
