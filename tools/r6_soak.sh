#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
mkdir -p gpurun_out
timeout -k 10 400 python tools/step_soak.py --seconds 60 --block 20 gpurun_out/r06_step_soak.txt 2>&1 | grep -v "^step \|psnr\|begin valid\|saved" | tail -20
STEPS=2000 timeout -k 10 400 python tools/e2e_driver.py 2>&1 | grep "DRIVER LOOP" | tee gpurun_out/r06_e2e_driver.txt
