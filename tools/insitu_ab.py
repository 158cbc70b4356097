"""Same-box, interleaved A/B of the timed step (bench.py, two-stream hipGraph) between kernel configurations: plan tables and/or
library builds.  The plan table is tuned on ISOLATED launches; three times in round 3 the isolated ranking of two conv variants
inverted under the two-stream schedule, so a variant is only adopted after this comparison.

    python tools/insitu_ab.py [--reps 3] name=table.json[,lib.so] name2=...      ('-' = the committed table / the product library)

e.g.  python tools/insitu_ab.py committed=- new=gpurun_out/tuned_plans.json old_lib=-,reconvat_amd/libreconvat_hip_old.so
Prints every run and, per variant, min / median ms per step.  Each run is a fresh child process (bench.py without its roofline,
parity and CPU legs).
"""
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def measure(variants, reps=3, verbose=True):
    """variants: [(name, table path or '-', library path or '-')]; returns {name: [ms per step of every run]} (runs interleaved)."""
    times = {v[0]: [] for v in variants}
    for _ in range(reps):
        for name, table, lib in variants:
            env = dict(os.environ)
            if table != '-':
                env['RV_PLAN_FILE'] = os.path.abspath(table)
            if lib != '-':
                env['RECONVAT_HIP_LIB'] = os.path.abspath(lib)
            env.pop('RV_AUTOTUNE', None)
            r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--no-parity', '--no-roofline', '--no-cpu-baseline'],
                               capture_output=True, text=True, env=env, cwd=ROOT)
            if r.returncode != 0:
                raise SystemExit(f'{name}: bench.py failed\n{r.stderr[-2000:]}')
            ms = json.loads(r.stdout.strip().split('\n')[-1])['mean_ms_per_step']
            times[name].append(ms)
            if verbose:
                print(f'{name:24s} {ms:.3f} ms', flush=True)
    return times


def main():
    args = sys.argv[1:]
    reps = 3
    if args and args[0] == '--reps':
        reps, args = int(args[1]), args[2:]
    variants = []
    for a in args:
        name, spec = a.split('=', 1)
        parts = spec.split(',')
        variants.append((name, parts[0], parts[1] if len(parts) > 1 else '-'))
    if len(variants) < 2:
        raise SystemExit(__doc__)
    times = measure(variants, reps)
    print()
    for name, ts in times.items():
        print(f'{name:24s} min {min(ts):.3f}  median {statistics.median(ts):.3f}  ({len(ts)} runs)')


if __name__ == '__main__':
    main()
