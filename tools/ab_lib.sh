#!/bin/bash
# Same-box A/B of a variant library (tools/build_variant.sh <name> -D...) against the product build: the training step,
# three alternating rounds, then the two-chain epilogue links.   usage: tools/ab_lib.sh <name>
set -euo pipefail
cd "$(dirname "$0")/.."
V=tools/_diag/$1.so
[ -f "$V" ] || { echo "missing $V: tools/build_variant.sh $1 -D..."; exit 1; }
[ "$V" -nt larvanet_amd/csrc/conv3x3_mfma.hip ] || { echo "$V is older than conv3x3_mfma.hip: rebuild the variant"; exit 1; }
step() { python bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms  %.1f Mpix/s   chain40 %.2f us  steady %.2f us  infer %.4f ms' % (d['ms_per_step'], d['value'], d['roofline']['avg_ms']*1e3, d['roofline']['avg_ms_steady_state']*1e3, d['infer']['ms_per_batch']))"; }
for round in 1 2 3; do
  echo "round $round  [$1]       $(LARVA_HIP_LIB=$V step)"
  echo "round $round  [product]  $(step)"
done
echo "--- two-chain epilogue links: $1, then product"
LARVA_HIP_LIB=$V python tools/bench_epilogues.py 2>&1 | grep -v amdgpu
python tools/bench_epilogues.py 2>&1 | grep -v amdgpu
for lib in $V larvanet_amd/csrc/liblarva_hip.so; do echo "[$lib] $(LARVA_HIP_LIB=$lib python tools/time_regions.py 2>&1 | grep dual_chain=True)"; done
