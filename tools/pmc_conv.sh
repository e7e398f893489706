#!/bin/bash
# PMC passes (one counter set per pass) over `bench.py --roofline-only`: the fused conv+ReLU layer as one
# whole-batch launch (conv3x3_mfma_kernel<48,true,1>) and as two half-batch strip launches
# (conv3x3_mfma_strip_kernel<1>).  Output: gpurun_out/r02_pmc_conv/<set>/...
set -euo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r02_pmc_conv
mkdir -p $O
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pass$i -- python3 $R/bench.py --roofline-only > $O/pass$i.log 2>&1 || exit 1
done
echo done
