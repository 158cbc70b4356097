#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python tools/pair_probe.py set3 > gpurun_out/r06_pair_probe3.txt 2>&1; cat gpurun_out/r06_pair_probe3.txt | cut -c1-400
