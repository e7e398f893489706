#!/bin/bash
# Round 4 same-box A/B of the strip kernel's latency trims:
#   product  = epilogue scalars pinned in SGPRs + the tile table inline in the kernel arguments
#   baseline = tools/_diag/nopin.so (-DLARVA_PIN_SCALARS=0) with LARVA_INLINE_TABLE=0 (round 3's behaviour)
#   table    = product library with LARVA_INLINE_TABLE=0 (the pinning alone)
set -euo pipefail
cd "$(dirname "$0")/.."
V=tools/_diag/nopin.so
[ -f "$V" ] || { echo "missing $V: tools/build_variant.sh nopin -DLARVA_PIN_SCALARS=0"; exit 1; }
[ "$V" -nt larvanet_amd/csrc/conv3x3_mfma.hip ] || { echo "$V is older than conv3x3_mfma.hip: rebuild the variant"; exit 1; }
step() { python bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms  %.1f Mpix/s   chain40 %.2f us  steady %.2f us  infer %.4f ms' % (d['ms_per_step'], d['value'], d['roofline']['avg_ms']*1e3, d['roofline']['avg_ms_steady_state']*1e3, d['infer']['ms_per_batch']))"; }
for round in 1 2 3; do
  echo "round $round  [baseline: no pin, device table]  $(LARVA_HIP_LIB=$V LARVA_INLINE_TABLE=0 step)"
  echo "round $round  [pin, device table]               $(LARVA_INLINE_TABLE=0 step)"
  echo "round $round  [product: pin + inline table]     $(step)"
done
echo "--- two-chain epilogue links, baseline then product"
LARVA_HIP_LIB=$V LARVA_INLINE_TABLE=0 python tools/bench_epilogues.py 2>&1 | grep -v amdgpu
python tools/bench_epilogues.py 2>&1 | grep -v amdgpu
