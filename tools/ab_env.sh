#!/bin/bash
# Same-box A/B of environment variants of the training step, two alternating rounds each, plus the
# un-profiled region timings.  usage: tools/ab_env.sh "VAR=a" "VAR=b" ...
set -euo pipefail
for round in 1 2; do
  for v in "$@"; do
    out=$(env $v python bench.py --no-cpu-baseline --no-extras --steps 50 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms  infer %.4f ms' % (d['ms_per_step'], d['infer']['ms_per_batch']))")
    echo "round $round  [$v]  $out"
  done
done
for v in "$@"; do echo "[$v] $(env $v python tools/time_regions.py 2>&1 | grep dual_chain=True)"; done
