#!/bin/bash
set -euo pipefail
set -x
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r02_pmc
mkdir -p $O
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/wgrad_fetch -- python3 $R/tools/bench_wgrad.py 32 8 3 hot > $O/wgrad_fetch.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/wgrad_write -- python3 $R/tools/bench_wgrad.py 32 8 3 hot > $O/wgrad_write.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/wgrad_mfma -- python3 $R/tools/bench_wgrad.py 32 8 3 hot > $O/wgrad_mfma.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/step -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/step.log 2>&1
echo rc=$?
cd $R && python tools/step_timeline.py gpurun_out/r02_pmc/step > gpurun_out/r02_pmc/step_summary.txt 2>&1; tail -30 gpurun_out/r02_pmc/step_summary.txt
