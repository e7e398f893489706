#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests/test_dp_overlap_gpu.py tests/test_drivers_gpu.py -x -q -m gpu > gpurun_out/r6_dp_tests.log 2>&1; rc=$?
tail -15 gpurun_out/r6_dp_tests.log
exit $rc
