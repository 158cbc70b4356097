# Winograd F(2x2,3x3) cost probe (timing only): the shipped 3x3 kernel with 4 instead of 9 MFMA groups per pixel tile (RV_ABLATE=256)
# and with the input-transform adds on top (768), against the same ablation build unmodified (0) and the product library (prod).
for cfg in "c3 16 16 640 229" "c3 32 32 320 114" "c3 64 64 160 57" "c3 128 128 80 28" "c3 96 48 160 57"; do
  echo -n "prod     "; python tools/bench_conv.py fwd $cfg 30 2>&1 | grep fwd
  for abl in 0 256 768; do
    echo -n "abl=$abl  "; RECONVAT_HIP_LIB=reconvat_amd/libreconvat_hip_abl.so RV_ABLATE=$abl python tools/bench_conv.py fwd $cfg 30 2>&1 | grep fwd
  done
done
