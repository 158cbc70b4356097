#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "fused_skip or bn_lrelu" 2>&1 | tail -4
python -m pytest tests/test_model_gpu.py tests/test_stress_gpu.py tests/test_plans_gpu.py -m gpu -q -x 2>&1 | tail -3
REPS=4 bash tools/knob_ab.sh "RV_FUSE_SKIP=0" "RV_FUSE_SKIP=1" "RV_FUSE_SKIP=2" 2>&1 | tail -4
