"""Coarse timeline of ONE optimiser step from a rocprofv3 (rocpd sqlite) kernel trace: per 0.5 ms bucket, the busy share of every queue and the
kernel that took most of the bucket on it -- where in the step only one chain has work.

    python tools/stream_timeline.py <results.db> [--bucket-us 500]
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
bucket = int(sys.argv[sys.argv.index('--bucket-us') + 1]) * 1000 if '--bucket-us' in sys.argv else 500000
adam = [r[0] for r in db.execute("select end from kernels where name like 'adam_k%' order by end").fetchall()]
lo, hi = adam[-3], adam[-2]
rows = db.execute(f'select start, end, queue_id, name from kernels where start >= {lo} and end <= {hi} order by start').fetchall()
queues = sorted({r[2] for r in rows}, key=lambda q: -sum(r[1] - r[0] for r in rows if r[2] == q))
print(f'# one step: {(hi - lo) / 1e6:.3f} ms, {len(rows)} kernels; queues by busy time: {queues}; bucket {bucket / 1e3:.0f} us')
nb = (hi - lo + bucket - 1) // bucket
for b in range(nb):
    t0, t1 = lo + b * bucket, min(hi, lo + (b + 1) * bucket)
    cells = []
    for q in queues:
        busy, top = 0, {}
        for s, e, qq, name in rows:
            if qq != q or e <= t0 or s >= t1:
                continue
            d = min(e, t1) - max(s, t0)
            busy += d
            nm = name.split('<')[0].split('(')[0].replace('void ', '')[:18]
            top[nm] = top.get(nm, 0) + d
        cells.append(f'{100 * busy / (t1 - t0):3.0f}% {max(top, key=top.get) if top else "-":18s}')
    print(f'{(t0 - lo) / 1e6:6.2f} ms  ' + ' | '.join(cells))
