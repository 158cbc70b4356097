"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average, like --stats.

    python tools/rocpd_stats.py gpurun_out/prof1/r1_results.db [--steps N] [--tail-ms X] > profiles/xyz.txt
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(.*\)$', '', name)
    name = name.replace('void ', '')
    return name[:110]


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = int(sys.argv[sys.argv.index('--steps') + 1]) if '--steps' in sys.argv else None
    where = ''
    if '--tail-ms' in sys.argv:      # only the dispatches of the last X ms of the trace (the timed steps)
        tail = float(sys.argv[sys.argv.index('--tail-ms') + 1])
        end = db.execute('select max(end) from kernels').fetchone()[0]
        where = f'where start >= {end - int(tail * 1e6)} '
    rows = db.execute('select name, count(*), sum(duration), avg(duration), min(duration), max(duration) '
                      f'from kernels {where}group by name order by sum(duration) desc').fetchall()
    total = sum(r[2] for r in rows)
    span = db.execute(f'select min(start), max(end) from kernels {where}').fetchone()
    print(f'# kernels: {sum(r[1] for r in rows)} dispatches, {len(rows)} distinct, total kernel time {total / 1e6:.3f} ms, '
          f'trace span {(span[1] - span[0]) / 1e6:.3f} ms')
    if steps:
        print(f'# per step (/{steps}): {total / 1e6 / steps:.3f} ms kernel time')
    print(f'{"calls":>8s} {"total_ms":>10s} {"avg_us":>9s} {"min_us":>9s} {"max_us":>9s} {"pct":>6s}  name')
    for name, n, tot, avg, mn, mx in rows:
        print(f'{n:8d} {tot / 1e6:10.3f} {avg / 1e3:9.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {100 * tot / total:6.2f}  {short(name)}')


if __name__ == '__main__':
    main()
