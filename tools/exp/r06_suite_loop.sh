#!/bin/bash
# N consecutive full `pytest -m gpu` runs on one box: pass / fail line of every run + the per-run maxima of the measured VAT-term errors
# (tests/parity_tol.py log) relative to their tolerances -> gpurun_out/r06_gpu_suite_runs.log      bash tools/exp/r06_suite_loop.sh [N]
cd $GRAFT_REPO_ROOT
N=${1:-10}
out=gpurun_out/r06_gpu_suite_runs.log
echo "# $N consecutive full -m gpu runs on one MI355X box, final round-6 build (source digest $(python -c 'from reconvat_amd import _lib; print(_lib.source_digest())'), plan table $(python -c 'from reconvat_amd import plans; print(plans.digest())'))" > $out
for i in $(seq 1 $N); do
  rm -f gpurun_out/parity_errors.jsonl
  python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -1 > /tmp/suite_tail.txt
  python - >> $out <<PY
import json
rows = [json.loads(l) for l in open('gpurun_out/parity_errors.jsonl')]
worst = max(rows, key=lambda r: r['rel_err'] / r['tol'])
vat = [r for r in rows if r['ref_spread'] is not None]
print('run $i:', open('/tmp/suite_tail.txt').read().strip(), '| loss-term checks', len(rows), '| worst share of tolerance %.2f (%s %s %.2e of %.2e)' % (worst['rel_err'] / worst['tol'], worst['case'], worst['key'].split('/')[-1], worst['rel_err'], worst['tol']),
      '| worst VAT-term error / reference spread %.2f' % max(r['rel_err'] / r['ref_spread'] for r in vat))
PY
done
cat $out
