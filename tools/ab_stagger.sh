#!/bin/bash
# How does the training step depend on the phase the two half-batch chains start in?  A sleeping launch of d us in front of
# chain 1 (d > 0) or chain 0 (d < 0) at every fork (LARVA_CHAIN_STAGGER_US), same box, two alternating rounds.
# usage: tools/ab_stagger.sh [d ...]
set -euo pipefail
cd "$(dirname "$0")/.."
ds=("$@"); [ ${#ds[@]} -gt 0 ] || ds=(0 1 2 3 4 6 8 10 12 -2 -4 -6 -8)
for round in 1 2; do
  for d in "${ds[@]}"; do
    LARVA_CHAIN_STAGGER_US=$d python bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 10 2>/dev/null | tail -1 |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round $round  stagger %6s us   step %.4f ms   infer %.4f ms' % ('$d', d['ms_per_step'], d['infer']['ms_per_batch']))"
  done
done
