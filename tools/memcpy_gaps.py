"""What the non-kernel nodes of the step's hipGraph cost: memory copies / memsets of one optimiser step (rocprofv3 --kernel-trace --memory-copy-trace)
with the kernel that ran before and after each on the device timeline.   python tools/memcpy_gaps.py <results.db>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
views = [r[0] for r in db.execute("select name from sqlite_master where type in ('view','table')").fetchall()]
print('# views:', ', '.join(v for v in views if 'copy' in v.lower() or v in ('kernels',)))
adam = [r[0] for r in db.execute("select end from kernels where name like 'adam_k%' order by end").fetchall()]
lo, hi = adam[-6], adam[-1]
steps = 5
cv = next((v for v in views if v.lower() in ('memory_copies', 'memory_copy')), None)
if cv is None:
    print('no memory-copy view'); sys.exit(0)
cols = [r[1] for r in db.execute(f'pragma table_info({cv})').fetchall()]
print('# columns:', cols)
rows = db.execute(f'select start, end, name, size from {cv} where start >= {lo} and end <= {hi} order by start').fetchall() if 'size' in cols else \
    db.execute(f'select start, end, name from {cv} where start >= {lo} and end <= {hi} order by start').fetchall()
ks = db.execute(f'select start, end, name from kernels where start >= {lo} and end <= {hi} order by start').fetchall()
print(f'# {len(rows) / steps:.1f} copies per step, total {sum(r[1] - r[0] for r in rows) / steps / 1e3:.1f} us per step of copy time')
import bisect
starts = [k[0] for k in ks]
for r in rows[:len(rows) // steps]:
    i = bisect.bisect_left(starts, r[0])
    before = max((k for k in ks[max(0, i - 40):i] if k[1] <= r[0]), key=lambda k: k[1], default=None)
    after = ks[bisect.bisect_left(starts, r[1])] if bisect.bisect_left(starts, r[1]) < len(ks) else None
    print(f'{r[2][:28]:28s} {r[3] if len(r) > 3 else "":>8} B  dur {(r[1] - r[0]) / 1e3:7.1f} us   idle before {(r[0] - before[1]) / 1e3 if before else -1:7.1f} us ({before[2][:24] if before else "-"})   to next kernel {(after[0] - r[1]) / 1e3 if after else -1:7.1f} us ({after[2][:24] if after else "-"})')
