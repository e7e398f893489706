#!/bin/bash
# HBM-side traffic of wgrad_reduce_kernel (two separate counter passes, as the guide prescribes), 32 layers x 8 partial images
set -euo pipefail
set -x
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r02_pmc_reduce
mkdir -p $O
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/tools/bench_wgrad.py 32 8 3 hot > $O/fetch.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/tools/bench_wgrad.py 32 8 3 hot > $O/write.log 2>&1
echo rc=$?
cd $R && python - <<'PY'
import csv, glob, collections
for what in ("fetch", "write"):
    f = glob.glob("gpurun_out/r02_pmc_reduce/%s/*/*counter_collection.csv" % what)[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0][-50:], r["Counter_Name"])
        agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    for k, (n, v) in sorted(agg.items()):
        print(what, k, "launches", n, "mean per launch", v / n)
PY
