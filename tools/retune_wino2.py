"""Offer the software-pipelined Winograd kernel (conv_wino2.hip: algo families 0x8NM / 0x9NM / 0xBNM) to every 3x3 entry of the shipped
plan table and write a CANDIDATE table in which the entries it wins by more than --min-gain percent are swapped (isolated launches,
exactly as keyed: batch, pixel strides, fused statistics / BatchNorm-backward reduction; operands rotated through scratch sets so that
no launch finds its input warm).  The candidate is adopted only after the same-box interleaved step A/B (tools/insitu_ab.py).

    python tools/retune_wino2.py [--out gpurun_out/plans_wino2.json] [--min-gain 2.0]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from reconvat_amd import ops, plans, _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'plans_wino2.json'))
ap.add_argument('--min-gain', type=float, default=2.0)
args = ap.parse_args()
dev = torch.device('cuda:0')
lib = _lib.load()
st = torch.cuda.current_stream()
NSETS = 3


def candidates(h, w, cout):
    ntile_n = (cout + 15) // 16
    wt_ = (w + 1) // 2
    out = []
    for fam, nw, tiles in ((8, 8, ((1, 1),)), (9, 8, ((1, 1), (2, 1), (1, 2))), (11, 4, ((1, 2), (2, 1)))):
        for nt, mt in tiles:
            if ntile_n % nt:
                continue
            th_max = min(h, 2 * ((nw * mt * 16) // wt_))
            ths = [0] + [t for t in range(2, th_max, 2) if -(-h // t) != -(-h // (t + 2)) and t >= th_max // 2]
            out += [t << 12 | fam << 8 | nt << 4 | mt for t in ths]
    return out


def timed(calls, algo):
    for c in calls:
        if c(algo) != 0:
            return None
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for i in range(6):
            calls[i % NSETS](algo)
        e1.record(st)
        e1.synchronize()
        t = e0.elapsed_time(e1) / 6 * 1e3
        best = t if best is None else min(best, t)
    return best


doc = json.load(open(plans.PLAN_FILE))
conv = dict(doc['conv'])
swapped, lines = 0, []
for k, algo in sorted(plans.conv_entries().items()):
    mode, bb, h, w, cin, cout, ild, old, stats, bnbwd = k
    if mode != 0 or cin % 16:
        continue
    calls = []
    keep = []
    wt = (torch.rand(cout, cin, 3, 3, device=dev) - 0.5) * 0.1
    bias = torch.zeros(cout, device=dev)
    wp = ops._pack('c3', wt, 'fwd')
    for s in range(NSETS):
        xbuf = torch.rand(bb, h, w, ild, device=dev) - 0.5
        obuf = torch.empty(bb, h, w, old, device=dev)
        ws = torch.zeros(ops.bn_ws_doubles(cout), device=dev, dtype=torch.float64) if stats else None
        z = coef = None
        if bnbwd:
            z = torch.rand(bb, h, w, cout, device=dev)
            coef = torch.rand(5 * cout, device=dev)
        keep.append((xbuf, obuf, ws, z, coef))
        a = (0, ops.ptr(xbuf), ild, bb, h, w, cin, ops.ptr(obuf), old, h, w, cout, ops.ptr(wp), ops.ptr(bias), 0)
        tail = (ops.ptr(ws) if ws is not None else None, ops.ptr(z) if z is not None else None, cout if z is not None else 0,
                ops.ptr(coef) if coef is not None else None, 0.01)
        calls.append(lambda al, a=a, tail=tail: lib.rv_conv_fwd(*a, al, *tail, st.cuda_stream))
    t0 = timed(calls, algo)
    res = sorted((t, c) for c in candidates(h, w, cout) for t in [timed(calls, c)] if t is not None)
    line = f'{",".join(str(int(x)) for x in k):<40} table {algo:#7x} {t0:6.1f} us'
    if res:
        t1, c1 = res[0]
        line += f'   pipelined best {c1:#7x} {t1:6.1f} us  x{t0 / t1:.3f}'
        if t1 < t0 * (1 - args.min_gain / 100):
            conv[','.join(str(int(x)) for x in k)] = int(c1)
            swapped += 1
            line += '  <- swapped'
    print(line, flush=True)
    lines.append(line)
    del keep, calls
doc['conv'] = conv
doc.setdefault('meta', {})['wino2'] = f'{swapped} 3x3 entries moved to the software-pipelined Winograd kernel (tools/retune_wino2.py, min gain {args.min_gain} %)'
with open(args.out, 'w') as fh:
    json.dump(doc, fh, indent=0, sort_keys=True)
    fh.write('\n')
with open(os.path.splitext(args.out)[0] + '_sweep.txt', 'w') as fh:
    fh.write('\n'.join(lines) + '\n')
print(f'{swapped} entries swapped -> {args.out}')
