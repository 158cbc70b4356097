#!/bin/bash
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench_line.err; tail -c 300 gpurun_out/r06_bench_line.json
bash tools/exp/r06_suite_loop.sh 4 > /dev/null 2>&1; cat gpurun_out/r06_gpu_suite_runs.log | cut -c1-250
python bench.py --model unet --batch-l 1 --no-cpu-baseline --no-roofline --no-parity | tail -1 | cut -c1-300
python tools/bench_onf.py 2>&1 | tail -2 | cut -c1-300
