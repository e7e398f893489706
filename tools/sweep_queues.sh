#!/bin/bash
set -euo pipefail
for v in "LARVA_DUAL_CHAIN=0" "LARVA_DUAL_CHAIN=1" "LARVA_DUAL_CHAIN=1 GPU_MAX_HW_QUEUES=8" "LARVA_DUAL_CHAIN=1 GPU_MAX_HW_QUEUES=2" "LARVA_DUAL_CHAIN=1 GPU_MAX_HW_QUEUES=16" "LARVA_DUAL_CHAIN=1 HIP_FORCE_DEV_KERNARG=1" ; do
  out=$(env $v python bench.py --no-cpu-baseline --no-extras --steps 50 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms  %.1f Mpix/s  infer %.3f ms' % (d['ms_per_step'], d['value'], d['infer']['ms_per_batch']))")
  echo "[$v]  $out"
done
