#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
mkdir -p gpurun_out
timeout -k 10 900 python tools/fuzz_medium.py > gpurun_out/r06_fuzz_medium.txt 2>&1; rc=$?
tail -25 gpurun_out/r06_fuzz_medium.txt
exit $rc
