#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_model_gpu.py -m gpu -q -k trajectory -s > gpurun_out/r06_traj.log 2>&1; grep -v Warning gpurun_out/r06_traj.log | grep "passed\|failed\|^onset_\|^frame_\|Error" | cut -c1-500 | tail -14
python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "winograd" > gpurun_out/r06_wino_tests.log 2>&1; tail -2 gpurun_out/r06_wino_tests.log
for i in 1 2; do timeout 600 python tools/pair_probe.py set4 2>&1 | grep -v amdgpu | cut -c1-420; done > gpurun_out/r06_pair_probe4.txt; cat gpurun_out/r06_pair_probe4.txt
