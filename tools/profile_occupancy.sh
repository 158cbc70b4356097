#!/bin/bash
# rocprofv3 kernel trace of the default bench command -> per-queue occupancy summary (tools/stream_occupancy.py)
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/occ_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --no-parity --steps 20 --warmup 5 "$@" > $out/bench.json 2> $out/bench.err
db=$(find $out -name "*results.db" | head -1)
cd $GRAFT_REPO_ROOT
python3 tools/stream_occupancy.py $db --last-steps 15 --pairs > gpurun_out/occ_${tag}.txt
sqlite3 $db "pragma table_info(kernels)" > gpurun_out/occ_${tag}_cols.txt 2>/dev/null
rm -rf $out/*/
