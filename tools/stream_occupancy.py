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

# ---- round 6: WHICH kernels co-run.  Per kernel name: its total time and the part of it during which at least one other kernel was in
# flight; and the heaviest co-running (name, name) pairs.  (the second chain only co-runs 9-14 % of the time; this
# table shows with whom -- a persistent conv workgroup takes a CU's LDS and most of its VGPRs, so what can share a CU with it is decided
# by the registers it leaves free.)
if '--pairs' in sys.argv:
    import re

    def short(nm):
        nm = re.sub(r'\(.*\)$', '', nm).replace('void ', '')
        return nm[:72]
    evs = []
    for i, (s, e, _, nm) in enumerate(rows):
        evs.append((s, 1, i)); evs.append((e, 0, i))
    evs.sort()
    live, last = set(), lo
    tot, shared, pairs = {}, {}, {}
    for t, kind, i in evs:
        dt = t - last
        if dt > 0 and live:
            names = sorted(short(rows[j][3]) for j in live)
            for nm in names:
                tot[nm] = tot.get(nm, 0) + dt
                if len(live) > 1:
                    shared[nm] = shared.get(nm, 0) + dt
            if len(live) == 2:
                pairs[tuple(names)] = pairs.get(tuple(names), 0) + dt
        last = t
        if kind:
            live.add(i)
        else:
            live.discard(i)
    print('\n# per kernel: ms/step in flight, of which co-running with another kernel')
    for nm, v in sorted(tot.items(), key=lambda kv: -kv[1])[:40]:
        print(f'{v / 1e6 / n:8.3f} ms/step  co-running {shared.get(nm, 0) / 1e6 / n:7.3f} ms ({100 * shared.get(nm, 0) / v:5.1f} %)  {nm}')
    print('\n# heaviest co-running pairs (exactly two kernels in flight)')
    for (a, b), v in sorted(pairs.items(), key=lambda kv: -kv[1])[:30]:
        print(f'{v / 1e6 / n:8.3f} ms/step  {a}  ||  {b}')
