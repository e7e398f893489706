#!/bin/bash
# Same-box A/B: LDS-DMA ring of two stages with three workgroups per CU (-DLARVA_RING_STAGES=2 -DLARVA_WG_PER_CU=3)
# against the product (three stages, two per CU): correctness of the variant first, then the batched launch (work queue
# always full: no launch-boundary effects), the chain schedules of tools/bench_dual_chain.py, and the step.
set -euo pipefail
cd "$(dirname "$0")/.."
V=tools/_diag/ring2.so
[ -f "$V" ] || { echo "missing $V: build it first (tools/build_variant.sh)"; exit 1; }
[ "$V" -nt larvanet_amd/csrc/conv3x3_mfma.hip ] || { echo "$V is older than conv3x3_mfma.hip: rebuild the variant"; exit 1; }
LARVA_HIP_LIB=$V python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "conv3x3 or strip or exits or batched" 2>&1 | tail -2
for lib in larvanet_amd/csrc/liblarva_hip.so $V; do
  echo "== $lib"
  LARVA_HIP_LIB=$lib python tools/bench_conv_batch.py 2>&1 | grep -v amdgpu
  LARVA_HIP_LIB=$lib python tools/bench_dual_chain.py 2>&1 | grep -E "one chain, 3x48|two half-batch chains, strip|complementary|three chains|four quarter" 
done
for round in 1 2; do
  for lib in larvanet_amd/csrc/liblarva_hip.so $V; do
    out=$(LARVA_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms  %.1f Mpix/s' % (d['ms_per_step'], d['value']))")
    echo "step, round $round  [$lib]  $out"
  done
done
