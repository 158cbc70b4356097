#!/bin/bash
cd $GRAFT_REPO_ROOT
REPS=2 bash tools/knob_ab.sh "-" "DEBUG_HIP_FORCE_GRAPH_QUEUES=1" "DEBUG_HIP_FORCE_GRAPH_QUEUES=2" "DEBUG_HIP_FORCE_GRAPH_QUEUES=3" "DEBUG_HIP_FORCE_GRAPH_QUEUES=4" "DEBUG_HIP_FORCE_GRAPH_QUEUES=8" \
   "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "DEBUG_HIP_GRAPH_BATCH_SIZE=1" "DEBUG_HIP_GRAPH_BATCH_SIZE=16" "DEBUG_HIP_GRAPH_BATCH_SIZE=256" > gpurun_out/r06_graph_knobs.txt 2>&1
cat gpurun_out/r06_graph_knobs.txt
for q in 1 2 4; do echo "== pair probe, DEBUG_HIP_FORCE_GRAPH_QUEUES=$q"; DEBUG_HIP_FORCE_GRAPH_QUEUES=$q timeout 300 python tools/pair_probe.py conv:64:64:160:57:911+bn:64:160:57 conv:64:64:160:57:911+attn 2>&1 | grep -v amdgpu | cut -c1-400; done > gpurun_out/r06_pair_probe_queues.txt 2>&1
cat gpurun_out/r06_pair_probe_queues.txt
