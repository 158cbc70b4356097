#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do python -m pytest tests/test_model_gpu.py -m gpu -q -k trajectory 2>&1 | tail -1; done
python bench.py > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench_line.err; tail -c 400 gpurun_out/r06_bench_line.json
