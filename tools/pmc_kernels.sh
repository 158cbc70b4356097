#!/bin/bash
# SQ counter passes (rocprofv3 --pmc, own runs) over single-kernel micro-benchmarks (tools/bench_conv.py).
#   bash tools/pmc_kernels.sh <outdir> "<bench_conv args>" ["<bench_conv args>" ...]
# Pass A: issue/wait split + MFMA busy.  Pass B: LDS bank conflicts / instruction mix.
set -u
out=$1; shift
mkdir -p $GRAFT_REPO_ROOT/$out
cd /tmp && export TMPDIR=/tmp
i=0
for cfg in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE \
      --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/a$i -- python3 $GRAFT_REPO_ROOT/tools/bench_conv.py $cfg 8 > $GRAFT_REPO_ROOT/$out/a$i.log 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM \
      --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/b$i -- python3 $GRAFT_REPO_ROOT/tools/bench_conv.py $cfg 8 > $GRAFT_REPO_ROOT/$out/b$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
for d in $out/a* $out/b*; do [ -d $d ] && python3 tools/pmc_summary.py $d mfma_k; [ -d $d ] && python3 tools/pmc_summary.py $d lds_k; [ -d $d ] && python3 tools/pmc_summary.py $d wino; done > $out/summary.txt 2>&1
# keep only the summaries (the raw csv files are large)
find $out -name "*.csv" -size +2M -delete
