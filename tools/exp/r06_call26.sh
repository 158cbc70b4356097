#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_stress_gpu.py -m gpu -q -x 2>&1 | tail -2
for s in "fwd c3 1 16 640 229" "fwd c3 16 1 640 229" "fwd c3 8 2 640 229" "fwd c3 2 8 640 229" "wgrad c3 1 16 640 229" "wgrad c3 8 2 640 229"; do python tools/bench_conv.py $s 30 2>&1 | tail -1; done
REPS=3 bash tools/knob_ab.sh "-" 2>&1 | tail -2
