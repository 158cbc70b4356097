#!/bin/bash
# A/B of the direct persistent 3x3 kernel (conv3x3_lds_k) between two library builds on a few BASELINE shapes / forced tiles.
#   bash tools/ab_direct.sh reconvat_amd/libreconvat_hip_r3.so
old=$1
for cfg in "0x713 c3 64 128 80 28" "0x723 c3 32 64 160 57" "0x713 c3 48 32 160 57" "0x412 c3 16 16 640 229" "0x8322 c3 192 96 80 28" "0x723 c3 32 32 320 114"; do
  set -- $cfg
  algo=$1; shift
  echo -n "$algo new: "; RV_FORCE_ALGO=$algo python tools/bench_conv.py fwd "$@" 30 2>&1 | grep fwd
  echo -n "$algo old: "; RECONVAT_HIP_LIB=$old RV_FORCE_ALGO=$algo python tools/bench_conv.py fwd "$@" 30 2>&1 | grep fwd
done
