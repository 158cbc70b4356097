"""Aggregate the [conv] lines of `bench.py --verbose` by launch class.   python tools/conv_classes.py gpurun_out/x.log [N top rows]"""
import collections
import re
import sys

agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
rows = []
for line in open(sys.argv[1]):
    m = re.match(r'\[conv\]\s+([\d.]+) ms/step\s+x\s+(\d+)\s+([\d.]+) ms\s+([\d.]+) TF/s\s+\((.*)\)', line)
    if not m:
        continue
    ms, cnt, each, tf, sig = float(m.group(1)), int(m.group(2)), float(m.group(3)), float(m.group(4)), m.group(5)
    f = [x.strip(" '") for x in sig.split(',')]
    key = (f[0], f[1], f[-1] if f[0] == 'rv_conv_fwd' else '')
    agg[key][0] += cnt
    agg[key][1] += ms
    agg[key][2] += ms * tf
    if f[0] == 'rv_conv_fwd':
        f[-2] = hex(int(f[-2]))
    rows.append((ms, cnt, each, tf, f))
tot = sum(v[1] for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f'{v[1]:7.3f} ms {100 * v[1] / tot:5.1f}%  n={v[0]:4d}  avgTF={v[2] / max(v[1], 1e-9):6.1f}  {k}')
print(f'{tot:.3f} ms total')
rows.sort(key=lambda r: -r[0])
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 0]:
    print(f'{r[0]:.3f} x{r[1]} {r[2] * 1e3:.1f}us {r[3]:.1f}TF', ' '.join(r[4]))
