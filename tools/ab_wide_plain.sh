#!/bin/bash
# Round 4: the 48-column tiles' mode-0 output with plain instead of non-temporal stores (-DLARVA_WIDE_PLAIN=1): the
# batched exits of the step, the full-image inference forward and the one-chain layer, same box.
set -euo pipefail
cd "$(dirname "$0")/.."
V=tools/_diag/wideplain.so
[ -f "$V" ] || { echo "missing $V"; exit 1; }
one() { python bench.py --no-cpu-baseline --steps 50 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); f=d['infer_full_image']; print('step %.4f ms  infer %.4f ms  full image V1 %.3f V2 %.3f V2-64 %.3f ms  single chain %.2f us' % (d['ms_per_step'], d['infer']['ms_per_batch'], f['LarvaNet']['ms_per_image'], f['LarvaNetV2']['ms_per_image'], f['LarvaNetV2_64ch']['ms_per_image'], d['roofline_single_chain']['avg_ms']*1e3))"; }
for round in 1 2; do
  echo "round $round  [wideplain] $(LARVA_HIP_LIB=$V one)"
  echo "round $round  [product]   $(one)"
done
