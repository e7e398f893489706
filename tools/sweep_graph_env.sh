#!/bin/bash
# HIP runtime switches against the cost of a launch inside a captured graph (tools/bench_chain_split.py, 32 and 48 channels)
set -uo pipefail
for v in "X=1" "DEBUG_HIP_FORCE_GRAPH_QUEUES=1" "DEBUG_HIP_FORCE_GRAPH_QUEUES=2" "DEBUG_HIP_FORCE_GRAPH_QUEUES=8" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" \
         "AMD_OPT_FLUSH=0" "DEBUG_HIP_GRAPH_BATCH_SIZE=1" "DEBUG_HIP_GRAPH_BATCH_SIZE=64" "DEBUG_HIP_DYNAMIC_QUEUES=1" "DEBUG_HIP_DYNAMIC_QUEUES=0" "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=2" "ROC_SYSTEM_SCOPE_SIGNAL=0" \
         "HIP_FORCE_DEV_KERNARG=0" "DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0" "GPU_STREAMOPS_CP_WAIT=0" "AMD_DIRECT_DISPATCH=0"; do
  echo "[$v]"
  env $v timeout -k 10 120 python tools/bench_chain_split.py 32 48 2>&1 | grep channels | sed 's/us per full-batch layer (fraction of 157.3 TFLOP.s)://'
done
