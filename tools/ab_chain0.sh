#!/bin/bash
# Round 4: chain 0 on the capture's own stream (LARVA_CHAIN0_ON_MAIN=1) against both chains on side streams (=0), same box.
set -euo pipefail
cd "$(dirname "$0")/.."
step() { python bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms  %.1f Mpix/s   infer %.4f ms' % (d['ms_per_step'], d['value'], d['infer']['ms_per_batch']))"; }
for round in 1 2 3; do
  for v in 0 1; do
    echo "round $round  [LARVA_CHAIN0_ON_MAIN=$v]  $(LARVA_CHAIN0_ON_MAIN=$v step)"
  done
done
for v in 0 1; do LARVA_CHAIN0_ON_MAIN=$v python tools/step_marks.py gpurun_out/r4_step_marks_chain0_$v.txt > /dev/null 2>&1; tail -4 gpurun_out/r4_step_marks_chain0_$v.txt; done
