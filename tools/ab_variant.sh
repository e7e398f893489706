#!/bin/bash
# Same-box A/B of a build variant (tools/build_variant.sh <name> -D...) against the product build:
#   tools/ab_variant.sh <name>     -> two-chain epilogue chains + the training step, alternating
set -euo pipefail
cd "$(dirname "$0")/.."
V=tools/_diag/$1.so
[ -f "$V" ] || { echo "missing $V: build it first (tools/build_variant.sh)"; exit 1; }
[ "$V" -nt larvanet_amd/csrc/conv3x3_mfma.hip ] || { echo "$V is older than conv3x3_mfma.hip: rebuild the variant"; exit 1; }
for round in 1 2; do
  python tools/bench_epilogues.py 2>&1 | grep -v amdgpu
  LARVA_HIP_LIB=$V python tools/bench_epilogues.py 2>&1 | grep -v amdgpu
done
for round in 1 2 3; do
  for lib in larvanet_amd/csrc/liblarva_hip.so $V; do
    out=$(LARVA_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms  %.1f Mpix/s' % (d['ms_per_step'], d['value']))")
    echo "step, round $round  [$lib]  $out"
  done
done
