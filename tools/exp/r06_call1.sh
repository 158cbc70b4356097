#!/bin/bash
# round 6, GPU call 1: trajectory test; co-running pair table of the shipped schedule; in-situ A/B of the low-VGPR table
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_model_gpu.py -m gpu -x -q -k trajectory > gpurun_out/r06_traj.log 2>&1; tail -15 gpurun_out/r06_traj.log
bash tools/profile_occupancy.sh r06a
head -60 gpurun_out/occ_r06a.txt
python tools/insitu_ab.py --reps 3 shipped=-,- lowreg=tools/exp/plans_lowreg.json,- > gpurun_out/r06_ab_lowreg.txt 2>&1; tail -4 gpurun_out/r06_ab_lowreg.txt
