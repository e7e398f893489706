#!/bin/bash
# Same-box A/B of a variant library against the product build: the training step, alternating rounds.
#   tools/r6_ab.sh <name> [rounds]      (tools/_diag/<name>.so)
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
V=tools/_diag/$1.so
[ -f "$V" ] || { echo "missing $V"; exit 1; }
step() { python bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms (min %.4f max %.4f)  %.1f Mpix/s   chain40 %.2f us' % (d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'], d['value'], d['roofline']['avg_ms']*1e3))"; }
for round in $(seq 1 ${2:-3}); do
  echo "round $round  [$1]       $(LARVA_HIP_LIB=$V step)"
  echo "round $round  [product]  $(step)"
done
