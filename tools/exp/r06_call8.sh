#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_model_gpu.py -m gpu -q -k trajectory -s > gpurun_out/r06_traj.log 2>&1; grep -v Warning gpurun_out/r06_traj.log | grep "passed\|failed\|Error" | cut -c1-500 | tail -5
python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "winograd_vs_torch" > gpurun_out/r06_wino_tests.log 2>&1; tail -2 gpurun_out/r06_wino_tests.log
python tools/bench_streams.py 2>&1 | grep -v amdgpu | tee gpurun_out/r06_bench_streams.txt
FULL=1 python tools/bench_streams.py 2>&1 | grep -v amdgpu | tee gpurun_out/r06_bench_streams_full.txt
