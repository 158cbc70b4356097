"""Summarise rocprofv3 --pmc CSV output (--output-format csv): per kernel, mean counter value per launch.

    python tools/pmc_summary.py gpurun_out/pmcX [name-substring]
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def collect(root):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
        per_dispatch = defaultdict(float)
        meta = {}
        for row in csv.DictReader(open(f)):
            key = (f, row['Dispatch_Id'], row['Counter_Name'])
            per_dispatch[key] += float(row['Counter_Value'])
            meta[key] = row['Kernel_Name']
        for key, v in per_dispatch.items():
            acc[meta[key]][key[2]].append(v)
    return acc


def table(roots):
    """--table dirA dirB ...: one line per kernel over ALL passes -- launches, the per-launch counter means that matter, derived ratios
    (matrix-pipe busy fraction of the kernel's cycles, VALU / SALU / LDS instructions per MFMA, LDS bank-conflict share)."""
    import re
    merged = defaultdict(dict)
    for root in roots:
        for kern, ctrs in collect(root).items():
            for name, vals in ctrs.items():
                merged[kern][name] = (sum(vals) / len(vals), len(vals))
    rows = []
    for kern, c in merged.items():
        g = c.get('GRBM_GUI_ACTIVE', (0, 0))
        n = g[1] or next(iter(c.values()))[1]
        get = lambda k: c.get(k, (0.0, 0))[0]
        mf = get('SQ_INSTS_MFMA')
        busy = get('SQ_VALU_MFMA_BUSY_CYCLES') / 1024 / (g[0] / 8) if g[0] else 0.0
        rows.append((g[0] * n, kern, n, g[0] / 8, busy, mf, get('SQ_INSTS_VALU'), get('SQ_INSTS_SALU'), get('SQ_INSTS_LDS'),
                     get('SQ_LDS_BANK_CONFLICT'), get('SQ_LDS_IDX_ACTIVE'), get('SQ_INSTS_VMEM_RD'), get('SQ_INSTS_VMEM_WR')))
    rows.sort(reverse=True)
    print('# per kernel, means per launch; cycles = GRBM_GUI_ACTIVE / 8 XCDs; pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / cycles;')
    print('# VALU / SALU / LDS per MFMA = instruction-count ratios (wave instructions); conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE')
    print(f'{"launches":>8s} {"cycles":>9s} {"pipe busy":>9s} {"MFMA":>10s} {"VALU/MFMA":>9s} {"SALU/MFMA":>9s} {"LDS/MFMA":>8s} {"conflict":>8s} {"VMEM rd":>9s} {"VMEM wr":>9s}  kernel')
    for _, kern, n, cyc, busy, mf, valu, salu, lds, conf, idx, rd, wr in rows:
        name = re.sub(r'\(.*\)$', '', kern).replace('void ', '')[:90]
        r = (lambda v: f'{v / mf:9.2f}') if mf else (lambda v: f'{"-":>9s}')
        print(f'{n:8d} {cyc:9.0f} {busy:9.3f} {mf:10.0f} {r(valu)} {r(salu)} {r(lds)[1:]} {(conf / idx if idx else 0):8.3f} {rd:9.0f} {wr:9.0f}  {name}')


def main():
    if sys.argv[1] == '--table':
        return table(sys.argv[2:])
    root = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
        per_dispatch = defaultdict(float)
        meta = {}
        for row in csv.DictReader(open(f)):
            key = (f, row['Dispatch_Id'], row['Counter_Name'])
            per_dispatch[key] += float(row['Counter_Value'])
            meta[key] = row['Kernel_Name']
        for key, v in per_dispatch.items():
            acc[meta[key]][key[2]].append(v)
    for kern, ctrs in sorted(acc.items()):
        if flt not in kern:
            continue
        print('==', kern[:120])
        for name, vals in sorted(ctrs.items()):
            print(f'    {name:28s} launches={len(vals):4d} mean={sum(vals) / len(vals):.6g}')
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in ctrs and 'GRBM_GUI_ACTIVE' in ctrs:
            m = sum(ctrs['SQ_VALU_MFMA_BUSY_CYCLES']) / len(ctrs['SQ_VALU_MFMA_BUSY_CYCLES'])
            g = sum(ctrs['GRBM_GUI_ACTIVE']) / len(ctrs['GRBM_GUI_ACTIVE'])
            # MFMA busy is summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs
            print(f'    -> MFMA busy per SIMD / kernel cycles = {m / 1024 / (g / 8):.3f}')


if __name__ == '__main__':
    main()
