#!/bin/bash
# timing ablations of wgrad_mfma_k (RV_ABLATION build; kernel durations from rocprofv3 --kernel-trace --stats)
export RECONVAT_HIP_LIB=$GRAFT_REPO_ROOT/reconvat_amd/libreconvat_hip_abl.so
cd /tmp && export TMPDIR=/tmp
for cfg in "wgrad c3 32 32 320 114"; do
  for abl in 0 15 31 47 63 64 79 16 32; do
    rm -rf /tmp/abl_out
    RV_ABLATE=$abl rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl_out -- python3 $GRAFT_REPO_ROOT/tools/bench_conv.py $cfg 30 > /dev/null 2>&1
    f=$(find /tmp/abl_out -name "*kernel_stats.csv" | head -1)
    echo "RV_ABLATE=$abl $cfg :: $(python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'wgrad_mfma_k' in r['Name'] or 'wgrad_reduce' in r['Name']:
        print(r['Name'][5:22], 'calls', r['Calls'], 'avg_us', round(float(r['AverageNs'])/1e3,1), end=' | ')
")"
  done
done
