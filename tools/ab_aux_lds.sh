#!/bin/bash
# Same-box A/B of the strip kernel's epilogue operands: through LDS (loader wave, round 3: -DLARVA_AUX_LDS=1)
# against per-lane global loads in the prologue (the product build).
set -euo pipefail
cd "$(dirname "$0")/.."
tools/build_variant.sh aux_lds -DLARVA_AUX_LDS=1 > /dev/null
for round in 1 2; do
  python tools/bench_epilogues.py
  LARVA_HIP_LIB=tools/_diag/aux_lds.so python tools/bench_epilogues.py
done
python tools/bench_epilogues.py 32
LARVA_HIP_LIB=tools/_diag/aux_lds.so python tools/bench_epilogues.py 32
for round in 1 2 3; do
  for lib in larvanet_amd/csrc/liblarva_hip.so tools/_diag/aux_lds.so; do
    out=$(LARVA_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms  %.1f Mpix/s' % (d['ms_per_step'], d['value']))")
    echo "step, round $round  [$lib]  $out"
  done
done
