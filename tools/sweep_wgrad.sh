#!/bin/bash
# wgrad_mfma_k partitioning sweep: waves per workgroup (RV_WGRAD_NW) x workgroups on the chip (RV_WGRAD_WGS); kernel
# durations from rocprofv3 --kernel-trace --stats
cd /tmp && export TMPDIR=/tmp
for cfg in "wgrad c3 32 32 320 114" "wgrad c3 64 64 160 57" "wgrad c3 128 128 80 28" "wgrad c3 16 16 640 229" "wgrad c3 96 48 160 57"; do
  for knobs in "8 256" "4 256" "4 512" "8 512" "4 768" "4 1024"; do
    set -- $knobs
    rm -rf /tmp/sw_out
    RV_WGRAD_NW=$1 RV_WGRAD_WGS=$2 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sw_out -- python3 $GRAFT_REPO_ROOT/tools/bench_conv.py $cfg 30 > /dev/null 2>&1
    f=$(find /tmp/sw_out -name "*kernel_stats.csv" | head -1)
    echo "NW=$1 WGS=$2 $cfg :: $(python3 -c "
import csv
for r in csv.DictReader(open('$f')):
    if 'wgrad_mfma_k' in r['Name'] or 'wgrad_reduce' in r['Name']:
        print(r['Name'][5:18], round(float(r['AverageNs'])/1e3,1), end=' | ')
")"
  done
done
