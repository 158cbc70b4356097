#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "rank1 or bn_lrelu" 2>&1 | tail -3
python -m pytest tests/test_model_gpu.py tests/test_stress_gpu.py -m gpu -q -x 2>&1 | tail -3
REPS=4 bash tools/knob_ab.sh "RV_FUSE_SKIP1=0" "RV_FUSE_SKIP1=1" 2>&1 | tail -3
