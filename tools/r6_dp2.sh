#!/bin/bash
# Rehearsal of `bench.py --gpus 2` on the ONE GPU of the box: two ranks over gloo, both on GPU 0 (RCCL refuses two ranks per
# device).  Not a scaling number (the ranks share the GPU): the N > 1 code path of bench.py with the real kernels.
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
mkdir -p gpurun_out
LARVA_DIST_BACKEND=gloo LARVA_BENCH_FULL=gpurun_out/r06_bench_2rank_gloo_full.json timeout -k 10 500 python bench.py --gpus 2 --steps 10 --warmup 3 > gpurun_out/r06_bench_2rank_gloo.json 2> gpurun_out/r06_bench_2rank_gloo.err; rc=$?
echo "rc $rc, line $(wc -c < gpurun_out/r06_bench_2rank_gloo.json) bytes"
cat gpurun_out/r06_bench_2rank_gloo.json
[ $rc -eq 0 ] || tail -30 gpurun_out/r06_bench_2rank_gloo.err
exit $rc
