#!/bin/bash
# SQ counters of EVERY kernel of the step (counters for more than one kernel): two rocprofv3 --pmc passes over eager
# steps of the bench workload (own runs, --kernel-trace only), one table per kernel name with the derived ratios.
#   bash tools/pmc_step_kernels.sh <tag>      -> gpurun_out/pmc_<tag>_step_kernels.txt
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/pmck_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $out/a -- python3 $GRAFT_REPO_ROOT/bench.py --no-graph --single-stream --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-parity > $out/a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM \
    --kernel-trace --output-format csv -d $out/b -- python3 $GRAFT_REPO_ROOT/bench.py --no-graph --single-stream --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-parity > $out/b.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py --table $out/a $out/b > gpurun_out/pmc_${tag}_step_kernels.txt 2>&1
find $out -name "*.csv" -size +1M -delete
