#!/bin/bash
# The Winograd kernel's skeleton (ablation build, event timing; a launch-only kernel shows the host pacing floor of the loop):
# RV_ABLATE 64 return at entry, 128 return after the staging plan, 1024 return after unit 0's DMA has landed, 63 whole loop with barrier /
# staging / MFMAs / epilogue / transform / weight reads off, 4159 = 63 + no statistics tail.
export RECONVAT_HIP_LIB=reconvat_amd/libreconvat_hip_abl.so
export RV_FORCE_ALGO=${1:-0x611}
for cfg in "c3 64 64 160 57" "c3 16 16 640 229"; do
  for abl in 0 64 128 1024 63 4159 4096; do
    echo -n "abl=$abl  "; RV_ABLATE=$abl timeout 60 python tools/bench_conv.py fwd $cfg 30 2>&1 | grep fwd
  done
done
