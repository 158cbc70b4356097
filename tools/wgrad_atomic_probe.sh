#!/bin/bash
# Cost probe (ablation build, wrong results by design): the row partitions of wgrad_mfma_k ADD their partial sums into ONE partial with fp32
# atomics (RV_ABLATE=2048) instead of storing one partial each -- what an in-kernel reduction into a per-layer staging buffer would issue.
export RECONVAT_HIP_LIB=reconvat_amd/libreconvat_hip_abl.so
for cfg in "c3 64 64 160 57" "c3 32 32 320 114" "c3 16 16 640 229" "c3 128 128 80 28" "c3 96 48 160 57" "c3 192 96 80 28"; do
  for abl in 0 2048 32; do
    echo -n "abl=$abl  "; RV_ABLATE=$abl python tools/bench_conv.py wgrad $cfg 30 2>&1 | grep wgrad
  done
done
