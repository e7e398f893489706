#!/bin/bash
# Round 6: the GPU suite (product tests), the measurement-library tests, then one bench run with --extras
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
mkdir -p gpurun_out
tag=${1:-a}
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r6_gpu_tests_$tag.log 2>&1; rc=$?
tail -5 gpurun_out/r6_gpu_tests_$tag.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python -m pytest tests -m diag -x -q > gpurun_out/r6_diag_tests_$tag.log 2>&1; rc=$?
tail -3 gpurun_out/r6_diag_tests_$tag.log
[ $rc -eq 0 ] || exit $rc
if [ "${2:-}" = "extras" ]; then
  LARVA_BENCH_FULL=gpurun_out/r6_bench_full_extras_$tag.json timeout -k 10 600 python bench.py --steps 20 --warmup 5 --extras > gpurun_out/r6_bench_extras_$tag.json 2> gpurun_out/r6_bench_extras_$tag.err || { tail -20 gpurun_out/r6_bench_extras_$tag.err; exit 1; }
  cat gpurun_out/r6_bench_extras_$tag.json
fi
