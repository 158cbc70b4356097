#!/bin/bash
# Ablation ladder of attn_fwd_k (VERDICT r04 item 3 iii): build attn.hip with -DRV_ATTN_ABL=mask for every mask (libraries next to the product one,
# wrong results by design), then time rv_local_attn_fwd alone per library (tools/attn_ablate.py).  Run `bash tools/attn_ablate.sh build` in the build
# container (hipcc), `bash tools/attn_ablate.sh run` on the MI355X.
cd "$(dirname "$0")/.."
masks="0 16 1 3 7 15 2 4 8"
if [ "$1" = build ]; then
  for m in $masks; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DRV_ATTN_ABL=$m -c reconvat_amd/csrc/attn.hip -o /tmp/attn_abl_$m.o || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls reconvat_amd/csrc/*.o | grep -v "attn.o\|conv_abl.o") /tmp/attn_abl_$m.o -o reconvat_amd/libreconvat_hip_attn$m.so || exit 1
  done
else
  for m in $masks; do RECONVAT_HIP_LIB=reconvat_amd/libreconvat_hip_attn$m.so python tools/attn_ablate.py $m; done
fi
