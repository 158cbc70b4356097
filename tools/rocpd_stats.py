"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average, like --stats, over a window of WHOLE
optimiser steps.

    python tools/rocpd_stats.py gpurun_out/prof1/r1_results.db [--last-steps N] [--json out.json] > profiles/xyz.txt

--json also writes the family sums as JSON; `conv_ms_per_step` (every conv* and wgrad* kernel: the four conv families) is what
bench.py's `roofline.frac_two_stream` divides the executed conv flops by when the trace is one of the shipped two-stream schedule.

The window is cut at optimiser-kernel (`adam_k`) boundaries: it runs from the end of the (N+1)-th-to-last `adam_k` to the end of
the last one, so it holds exactly N steps (N defaults to every complete step but the first five = warm-up) and "per step"
figures divide by the step count that is actually in the window.  A per-family table (conv3x3 / conv 1x1,2x2 / wgrad / gemm / attention /
BatchNorm / front-end / losses+VAT / optimiser) follows the per-kernel one.
"""
import re
import sqlite3
import sys

def _src_digest():
    """Digest of the kernel sources the profiled library was built from (reconvat_amd/build.py::source_digest)."""
    import os as _os
    sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
    from reconvat_amd import build as _b
    return _b.source_digest()


FAMILIES = [
    ('conv3x3 fwd+dgrad (MFMA)', ('conv3x3_lds_k', 'conv3x3_wino_k', 'conv3x3_wino2_k')),
    ('conv direct 3x3/1x1/2x2 (MFMA / HBM)', ('conv_mfma_k',)),
    ('conv C<=2 layers (HBM)', ('conv_small_k', 'wgrad_small_k', 'conv_narrow_out_k', 'conv_cin12_k', 'wgrad_small_sw_k', 'wgrad_cin1_k')),
    ('weight gradients (MFMA)', ('wgrad_mfma_k', 'wgrad_wino_k', 'wgrad_reduce', 'sums_fold')),
    ('linear GEMMs (MFMA)', ('gemm_', 'colsum', 'sigmoid', 'zero_strided')),
    ('local attention', ('attn_',)),
    ('BatchNorm + leaky-ReLU (HBM)', ('bn_',)),
    ('front-end (HBM)', ('mel_',)),
    ('losses / VAT elementwise (HBM)', ('reduce_', 'loss_bwd', 'vat_', 'elementwise', 'vectorized', 'Philox', 'distribution')),
    ('optimiser + weight repack', ('adam_k', 'pack_', 'clip', 'counter_add')),
]


def short(name):
    name = re.sub(r'\(.*\)$', '', name)
    name = name.replace('void ', '')
    return name[:110]


def main():
    db = sqlite3.connect(sys.argv[1])
    adam = [r[0] for r in db.execute("select end from kernels where name like 'adam_k%' order by end").fetchall()]
    if len(adam) < 2:
        raise SystemExit('fewer than two adam_k dispatches in the trace: cannot cut whole steps')
    n = int(sys.argv[sys.argv.index('--last-steps') + 1]) if '--last-steps' in sys.argv else max(1, len(adam) - 1 - 5)
    n = min(n, len(adam) - 1)
    lo, hi = adam[-1 - n], adam[-1]
    where = f'where start >= {lo} and end <= {hi} '
    rows = db.execute('select name, count(*), sum(duration), avg(duration), min(duration), max(duration) '
                      f'from kernels {where}group by name order by sum(duration) desc').fetchall()
    total = sum(r[2] for r in rows)
    steps = db.execute(f"select count(*) from kernels {where}and name like 'adam_k%'").fetchone()[0]
    print(f'# window: {steps} whole optimiser steps (cut at adam_k boundaries), wall {(hi - lo) / 1e6:.3f} ms = '
          f'{(hi - lo) / 1e6 / steps:.3f} ms/step')
    print(f'# kernels: {sum(r[1] for r in rows)} dispatches ({sum(r[1] for r in rows) / steps:.0f}/step), {len(rows)} distinct, '
          f'summed kernel time {total / 1e6:.3f} ms = {total / 1e6 / steps:.3f} ms/step')
    print(f'{"calls":>8s} {"calls/st":>8s} {"total_ms":>10s} {"ms/step":>8s} {"avg_us":>9s} {"min_us":>9s} {"max_us":>9s} {"pct":>6s}  name')
    for name, c, tot, avg, mn, mx in rows:
        print(f'{c:8d} {c / steps:8.1f} {tot / 1e6:10.3f} {tot / 1e6 / steps:8.3f} {avg / 1e3:9.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} '
              f'{100 * tot / total:6.2f}  {short(name)}')
    print('\n# by family (summed kernel time per step; concurrent chains overlap, so the sum exceeds the wall time)')
    fam = {f: [0, 0.0] for f, _ in FAMILIES}
    fam['other'] = [0, 0.0]
    for name, c, tot, *_ in rows:
        key = next((f for f, pats in FAMILIES if any(p in name for p in pats)), 'other')
        fam[key][0] += c
        fam[key][1] += tot
    for f, (c, tot) in fam.items():
        if c:
            print(f'{c / steps:8.1f} launches/step {tot / 1e6 / steps:8.3f} ms/step {100 * tot / total:6.2f} %  {f}')
    if '--json' in sys.argv:
        import json
        conv_fams = [f for f, _ in FAMILIES[:4]]
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from reconvat_amd import plans
        doc = {'steps': steps, 'wall_ms_per_step': (hi - lo) / 1e6 / steps, 'kernel_ms_per_step': total / 1e6 / steps,
               # the tile table (and commit) the traced command ran: bench.py only uses the conv time of a trace whose table is its own
               'kernel_plan_table': plans.digest(), 'source_digest': _src_digest(), 'git': os.environ.get('RV_GIT_SHA', 'not recorded'),
               'launches_per_step': sum(r[1] for r in rows) / steps,
               'conv_ms_per_step': sum(fam[f][1] for f in conv_fams) / 1e6 / steps,
               'conv_launches_per_step': sum(fam[f][0] for f in conv_fams) / steps,
               'families_ms_per_step': {f: tot / 1e6 / steps for f, (c, tot) in fam.items() if c},
               'winograd_ms_per_step': sum(r[2] for r in rows if 'conv3x3_wino' in r[0]) / 1e6 / steps}
        with open(sys.argv[sys.argv.index('--json') + 1], 'w') as fh:
            json.dump(doc, fh, indent=1)


if __name__ == '__main__':
    main()
