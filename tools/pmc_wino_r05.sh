export RV_FORCE_ALGO=0x811; bash tools/pmc_kernels.sh gpurun_out/pmc_w2 "fwd c3 64 64 160 57" "fwd c3 96 48 160 57" "fwd c3 16 16 640 229" > /dev/null 2>&1
export RV_FORCE_ALGO=0x611; bash tools/pmc_kernels.sh gpurun_out/pmc_w1 "fwd c3 64 64 160 57" "fwd c3 96 48 160 57" "fwd c3 16 16 640 229" > /dev/null 2>&1
unset RV_FORCE_ALGO
for d in gpurun_out/pmc_w2 gpurun_out/pmc_w1; do for i in 1 2 3; do python3 tools/pmc_summary.py --table $d/a$i $d/b$i | grep "wino\|launches"; done; done
