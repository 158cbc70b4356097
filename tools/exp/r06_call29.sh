#!/bin/bash
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench_line.err; tail -c 200 gpurun_out/r06_bench_line.json
bash tools/exp/r06_suite_loop.sh 3 > /dev/null 2>&1; cat gpurun_out/r06_gpu_suite_runs.log | cut -c1-200
