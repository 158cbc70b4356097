#!/bin/bash
# round 6: the measurement pass on the final build -- full GPU suite, smoke, kernel traces (two-stream, single-stream, occupancy), PMC passes
#   bash tools/exp/r06_final_measure.sh <git sha>
cd $GRAFT_REPO_ROOT
export RV_GIT_SHA=$1
rm -f gpurun_out/parity_errors.jsonl
python -m pytest tests -m gpu -q > gpurun_out/r06_suite_final.log 2>&1; tail -3 gpurun_out/r06_suite_final.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.txt 2>&1; tail -3 gpurun_out/r06_smoke.txt | cut -c1-300
bash tools/profile_step.sh r06
bash tools/profile_step.sh r06ss --single-stream
bash tools/profile_occupancy.sh r06f
bash tools/pmc_step_traffic.sh r06 $1
bash tools/pmc_step_kernels.sh r06
ls -la gpurun_out | grep -i "r06\|prof_\|pmc_\|occ_" | tail -20
