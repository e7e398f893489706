#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_hip_kernels.py -x -q -k "head" 2>&1 | tail -8
for v in ${1:-0 1}; do echo "== LARVA_HEAD_MFMA=$v"; LARVA_HEAD_MFMA=$v timeout -k 10 200 python tools/bench_head_bicubic.py 2>&1 | tail -2; done | tee gpurun_out/r6_head_ab.txt
