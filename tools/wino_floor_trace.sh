#!/bin/bash
# rocprofv3 kernel durations of the Winograd kernel's early-return probes (ablation build; see tools/wino_floor.sh for the bits)
export RECONVAT_HIP_LIB=$GRAFT_REPO_ROOT/reconvat_amd/libreconvat_hip_abl.so
export RV_FORCE_ALGO=${1:-0x611}
out=$GRAFT_REPO_ROOT/gpurun_out/wino_floor
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for abl in 0 64 128 1024 63; do
  export RV_ABLATE=$abl
  timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d $out/a$abl -- python3 $GRAFT_REPO_ROOT/tools/bench_conv.py fwd c3 64 64 160 57 30 > $out/a$abl.log 2>&1
  f=$(find $out/a$abl -name "*kernel_stats.csv" | head -1)
  echo "abl=$abl $(grep wino_k $f | sed 's/"[^"]*"/K/' | cut -d, -f1-4)"
done
