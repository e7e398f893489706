#!/bin/bash
# Matrix-pipe occupancy and clock of the (32, 32) flat weight-gradient launch: one rocprofv3 counter pass.
set -euo pipefail
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_pmc_c32
mkdir -p $O
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/mfma -- python3 $R/tools/bench_wgrad_widths.py 32 64 > $O/mfma.log 2>&1
cd $R && python - <<'PY'
import csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r04_pmc_c32/mfma/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/r04_pmc_c32/mfma/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, c in rows.items():
    if "wgrad3x3" not in k:
        continue
    n = len(c["GRBM_GUI_ACTIVE"])
    gui = sum(c["GRBM_GUI_ACTIVE"]) / n
    mf = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / n
    d = sum(dur[k]) / len(dur[k])
    print("%s: %d launches, %.1f us, clock %.3f GHz, MFMA busy %.1f %% of SIMD cycles" % (k, n, d, gui / 8 / d / 1e3, 100 * mf / 1024 / (gui / 8)))
PY
