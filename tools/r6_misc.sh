#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
O=gpurun_out/r06_prof
mkdir -p $O
python tools/diag_overlap.py $O/dual_chain_overlap.txt > $O/diag_overlap.log 2>&1 || { echo "diag_overlap failed"; tail -5 $O/diag_overlap.log; }
for i in 1 2 3; do python tools/diag_step.py $O/step_timeline_stamped_$i.txt > /dev/null 2>&1 || echo "diag_step $i failed"; grep -n "^forward chain\|^backward chain\|^stamps:" $O/step_timeline_stamped_$i.txt; done
tools/r6_bench3.sh b
