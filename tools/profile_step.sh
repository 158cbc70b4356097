#!/bin/bash
# rocprofv3 kernel trace of the default bench command -> per-kernel / per-family table over whole optimiser steps
#   bash tools/profile_step.sh <tag> [extra bench.py flags]
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --no-parity --steps 20 --warmup 5 "$@" > $out/bench.json 2> $out/bench.err
db=$(find $out -name "*results.db" | head -1)
cd $GRAFT_REPO_ROOT
python3 tools/rocpd_stats.py $db --last-steps 15 --json gpurun_out/prof_${tag}_stats.json > gpurun_out/prof_${tag}_stats.txt
rm -rf $out/*/   # keep the table and the bench line, drop the raw trace
