#!/bin/bash
# In-situ A/B of environment knobs: interleaved bench.py runs (fresh processes) per setting, ms per step, sorted per setting.
#   REPS=6 bash tools/knob_ab.sh "RV_BN_MAXBLK=1023" "RV_BN_MAXBLK=768" "RV_BN_MAXBLK=768 RV_BN_MAXBLK_BWD=639"      ("-" = no variable: the defaults)
cd "$(dirname "$0")/.."
for rep in $(seq 1 ${REPS:-4}); do
  for cfg in "$@"; do
    if [ "$cfg" = "-" ]; then envs=""; else envs="$cfg"; fi
    ms=$(env $envs python bench.py --no-cpu-baseline --no-roofline --no-parity 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['mean_ms_per_step'])")
    echo "[${cfg// /,}] $ms"
  done
done | sort | awk '{k=$1; a[k]=a[k]" "$2} END {for (k in a) print k, a[k]}' | sort
