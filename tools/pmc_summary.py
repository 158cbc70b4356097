"""Summarise rocprofv3 --pmc CSV output (--output-format csv): per kernel, mean counter value per launch.

    python tools/pmc_summary.py gpurun_out/pmcX [name-substring]
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
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
