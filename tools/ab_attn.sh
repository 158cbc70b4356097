#!/bin/bash
# A/B of the local-attention kernels between two library builds (tools/bench_attn.py: fwd + bwd incl. the projection GEMMs, HIP events).
#   bash tools/ab_attn.sh reconvat_amd/libreconvat_hip_r3.so
old=$1
for cfg in "768 6 176" "916 4 88" "916 4 229"; do
  for i in 1 2 3; do
    echo -n "new: "; timeout 120 python tools/bench_attn.py $cfg 2>&1 | grep attention
    echo -n "old: "; RECONVAT_HIP_LIB=$old timeout 120 python tools/bench_attn.py $cfg 2>&1 | grep attention
  done
done
