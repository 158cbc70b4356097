#!/bin/bash
# Cost probe for a Winograd F(3x3, 2x2) weight-gradient kernel, run BEFORE writing one (ablation build, wrong results by design):
# the direct kernel wgrad_mfma_k with 4 of its 9 MFMA groups and 4 + TB of its 9 + TB fragment reads per 4-pixel k-step -- what the
# Winograd form issues per 4 pixels (16 MFMA groups, 16 + 4 TB reads per 4 tiles = 16 pixels) --, then also with its transform adds
# (14 per k-step).  Everything else (staging, barriers, fold, partial-sum stores) unchanged; the 16/9 larger partial sums are not modelled.
export RECONVAT_HIP_LIB=reconvat_amd/libreconvat_hip_abl.so
for cfg in "c3 64 64 160 57" "c3 32 32 320 114" "c3 16 16 640 229" "c3 128 128 80 28" "c3 96 48 160 57" "c3 48 24 320 114" "c3 192 96 80 28"; do
  for abl in 0 256 768 1; do
    echo -n "abl=$abl  "; RV_ABLATE=$abl python tools/bench_conv.py wgrad $cfg 30 2>&1 | grep wgrad
  done
done
