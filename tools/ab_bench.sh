#!/bin/bash
# Same-box A/B of train-step variants (box-to-box spread is ~5 %, larger than most effects).
# usage: tools/ab_bench.sh "ENV1=a ENV2=b" "ENV1=c" ...   (each argument = one variant's environment)
set -euo pipefail
for round in 1 2; do
  for v in "$@"; do
    out=$(env $v python bench.py --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms  %.1f Mpix/s' % (d['ms_per_step'], d['value']))")
    echo "round $round  [$v]  $out"
  done
done
