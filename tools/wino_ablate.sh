# Where the Winograd kernel's time goes (timing only, ablation build): RV_ABLATE bits 1 barrier, 2 staging, 4 MFMAs, 8 epilogue,
# 16 patch reads + input transform, 32 weight-fragment reads.
export RECONVAT_HIP_LIB=reconvat_amd/libreconvat_hip_abl.so
export RV_FORCE_ALGO=${1:-0x611}
for cfg in "c3 64 64 160 57" "c3 16 16 640 229"; do
  for abl in 0 1 2 4 8 16 32 48 60 63 62; do
    echo -n "abl=$abl  "; RV_ABLATE=$abl python tools/bench_conv.py fwd $cfg 30 2>&1 | grep fwd
  done
done
