#!/bin/bash
# A/B of the head conv (LarvaHead, K = 27): direct kernel vs the padded-MFMA launch, same box.
set -euo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  LARVA_HEAD_DIRECT=$v LARVA_DUAL_CHAIN=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_head_$v -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2>&1
done
cd $R
for v in 1 0; do echo "== LARVA_HEAD_DIRECT=$v (single chain, rocprofv3 kernel stats)"; python tools/step_timeline.py gpurun_out/r02_head_$v | grep -E "kernels, span|head_conv3|mfma_kernel<48, true, 0>|elementwise|wgrad3x3_kernel<48, 16"; done
for v in 1 0; do echo "== LARVA_HEAD_DIRECT=$v un-profiled step"; LARVA_HEAD_DIRECT=$v python bench.py --no-cpu-baseline --no-extras --steps 50 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['infer'])"; done
