#!/bin/bash
# Round 4, VERDICT r3 item 1a: ReLU backward from sign bits against the fp32 mask operand, same box.
#   (1) the strip kernel's epilogues as two-chain links (tools/bench_epilogues.py: relu / relu+bits / mask / maskbits)
#   (2) the training step, LARVA_MASK_BITS=0 / 1 alternating, three rounds, + un-profiled region timings
set -euo pipefail
cd "$(dirname "$0")/.."
python tools/bench_epilogues.py 2>&1 | grep -v amdgpu
for round in 1 2 3; do
  for v in 0 1; do
    out=$(LARVA_MASK_BITS=$v python bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms  %.1f Mpix/s' % (d['ms_per_step'], d['value']))")
    echo "step, round $round  [LARVA_MASK_BITS=$v]  $out"
  done
done
for v in 0 1; do echo "[LARVA_MASK_BITS=$v] $(LARVA_MASK_BITS=$v python tools/time_regions.py 2>&1 | grep dual_chain=True)"; done
