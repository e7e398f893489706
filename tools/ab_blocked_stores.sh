#!/bin/bash
# DESIGN 9.0, priced before building it: the strip kernel's epilogue writing each workgroup's tile as ONE contiguous
# run (timing-only build, -DLARVA_DIAG=1024: results land in the wrong places; every load is left as it is) against
# the product build, same box: two half-batch chains, every link conv+ReLU / +res0 / mask, and the training step.
set -euo pipefail
cd "$(dirname "$0")/.."
tools/build_variant.sh blocked_stores -DLARVA_DIAG=1024 > /dev/null
for round in 1 2; do
  python tools/bench_epilogues.py
  LARVA_HIP_LIB=tools/_diag/blocked_stores.so python tools/bench_epilogues.py
done
for round in 1 2; do
  for lib in larvanet_amd/csrc/liblarva_hip.so tools/_diag/blocked_stores.so; do
    out=$(LARVA_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms  %.1f Mpix/s' % (d['ms_per_step'], d['value']))")
    echo "step, round $round  [$lib]  $out"
  done
done
