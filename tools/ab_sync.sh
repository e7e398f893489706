#!/bin/bash
# Same-box A/B of how `return loss.item()` waits: LARVA_EARLY_LOSS=poll | split | 0 (bench.py's value_sync_loss leg)
set -euo pipefail
for round in 1 2; do
  for v in poll split 0; do
    out=$(LARVA_EARLY_LOSS=$v python bench.py --no-cpu-baseline --steps 50 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('async %.4f ms  sync %.4f ms' % (d['ms_per_step'], d['value_sync_loss']['ms_per_step']))")
    echo "round $round  [LARVA_EARLY_LOSS=$v]  $out"
  done
done
