#!/bin/bash
# HBM traffic of the conv phase: separate rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE) over eager steps of the bench
# workload, summarised by tools/pmc_traffic.py.   bash tools/pmc_step_traffic.sh <tag> <git sha>
tag=$1; export RV_GIT_SHA=$2
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/$ctr -- python3 $GRAFT_REPO_ROOT/bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-parity > $out/$ctr.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 tools/pmc_traffic.py $out > gpurun_out/pmc_${tag}_traffic.json
find $out -name "*.csv" -size +1M -delete
