#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_model_gpu.py -m gpu -q -k trajectory -s > gpurun_out/r06_traj.log 2>&1; grep -v Warning gpurun_out/r06_traj.log | grep "passed\|failed\|onset_\|frame_\|Error" | cut -c1-400 | tail -20
timeout 600 python tools/pair_probe.py set2 > gpurun_out/r06_pair_probe2.txt 2>&1; cat gpurun_out/r06_pair_probe2.txt | cut -c1-260
