#!/bin/bash
# Round 6: the default bench three times back to back (run-to-run spread of roofline.frac, wall time per section)
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
mkdir -p gpurun_out
tag=${1:-a}
shift || true
for i in 1 2 3; do
  t0=$(date +%s.%N)
  LARVA_BENCH_FULL=gpurun_out/r6_bench_full_${tag}$i.json python bench.py --steps 20 --warmup 5 "$@" > gpurun_out/r6_bench_${tag}$i.json 2> gpurun_out/r6_bench_${tag}$i.err || { echo "run $i failed"; tail -20 gpurun_out/r6_bench_${tag}$i.err; exit 1; }
  t1=$(date +%s.%N)
  echo "run $i: wall $(python3 -c "print(round($t1 - $t0, 1))") s, line $(wc -c < gpurun_out/r6_bench_${tag}$i.json) bytes"
  cat gpurun_out/r6_bench_${tag}$i.json
done
