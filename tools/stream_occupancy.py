"""Per-queue timeline summary of a rocprofv3 (rocpd sqlite) kernel trace over whole optimiser steps: busy time of each HIP
stream / HSA queue, time with 0 / 1 / 2+ kernels in flight, and the launch gaps on the busiest queue.

    python tools/stream_occupancy.py gpurun_out/prof_x/<host>/<pid>_results.db [--last-steps N]
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
qcol = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else None)
adam = [r[0] for r in db.execute("select end from kernels where name like 'adam_k%' order by end").fetchall()]
n = int(sys.argv[sys.argv.index('--last-steps') + 1]) if '--last-steps' in sys.argv else max(1, len(adam) - 6)
lo, hi = adam[-1 - n], adam[-1]
rows = db.execute(f'select start, end, {qcol or 0}, name from kernels where start >= {lo} and end <= {hi} order by start').fetchall()
wall = (hi - lo) / 1e6 / n
print(f'# {n} steps, wall {wall:.3f} ms/step, {len(rows) / n:.0f} kernels/step, queue column: {qcol}')
by_q = {}
for s, e, q, _ in rows:
    by_q.setdefault(q, []).append((s, e))
for q, iv in sorted(by_q.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
    busy = sum(e - s for s, e in iv) / 1e6 / n
    gaps = [iv[i + 1][0] - iv[i][1] for i in range(len(iv) - 1)]
    small = [g for g in gaps if 0 <= g < 50000]
    print(f'queue {q}: {len(iv) / n:.0f} kernels/step, busy {busy:.3f} ms/step, '
          f'gaps < 50 us: {len(small) / n:.0f}/step, mean {sum(small) / max(1, len(small)) / 1e3:.2f} us, total {sum(small) / 1e6 / n:.3f} ms/step')
# concurrency histogram
ev = []
for s, e, _, _ in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth, last, hist = 0, lo, {}
for t, d in ev:
    hist[depth] = hist.get(depth, 0) + (t - last)
    last = t
    depth += d
hist[0] = hist.get(0, 0) + (hi - last)
for k in sorted(hist):
    print(f'{k} kernels in flight: {hist[k] / 1e6 / n:.3f} ms/step ({100 * hist[k] / (hi - lo):.1f} %)')
