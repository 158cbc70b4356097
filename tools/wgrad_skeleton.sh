#!/bin/bash
# Where the non-MFMA time of wgrad_mfma_k goes (ablation build; RV_ABLATE bits: 1 no MFMAs, 2 no fragment reads, 4 stage the first row only,
# 8 no row barriers, 16 no LDS zeroing, 32 return before the fold / partial-sum stores, 64 return at once)
export RECONVAT_HIP_LIB=reconvat_amd/libreconvat_hip_abl.so
for cfg in "c3 64 64 160 57" "c3 32 32 320 114"; do
  for abl in 0 1 3 7 15 31 63 64 2 4 8 16 32; do
    echo -n "abl=$abl  "; RV_ABLATE=$abl python tools/bench_conv.py wgrad $cfg 30 2>&1 | grep wgrad
  done
done
