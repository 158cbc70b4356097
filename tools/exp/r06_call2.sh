#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_model_gpu.py -m gpu -x -q -k trajectory -s > gpurun_out/r06_traj.log 2>&1; grep -v Warning gpurun_out/r06_traj.log | tail -15
timeout 600 python tools/pair_probe.py > gpurun_out/r06_pair_probe.txt 2>&1; cat gpurun_out/r06_pair_probe.txt | cut -c1-260
