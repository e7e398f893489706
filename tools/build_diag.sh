#!/bin/bash
# The measurement library: liblarva_hip.so's sources + the entry points that exist only under -DLARVA_DIAG_API (declared
# in tools/larva_diag.h: kernel-attached launch timing, the stamp / delay marker launches, the pair-chain probe), into
# tools/_diag/<name>.so.  conv3x3_mfma.hip and larva_pointwise.hip are recompiled (extra -D options go to both), the
# weight-gradient object is the product's (python -m larvanet_amd.build first).
#   tools/build_diag.sh diag [-DLARVA_DIAG_ONLY48=1 ...]
#   LARVA_HIP_LIB=tools/_diag/diag.so python tools/probe_pair_chain.py
set -euo pipefail
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/_diag
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -DLARVA_DIAG_API=1 -Ilarvanet_amd/csrc"
hipcc $FLAGS "$@" -c larvanet_amd/csrc/conv3x3_mfma.hip -o tools/_diag/$name.conv.o &
hipcc $FLAGS "$@" -c larvanet_amd/csrc/larva_pointwise.hip -o tools/_diag/$name.pointwise.o &
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_diag/$name.so tools/_diag/$name.conv.o tools/_diag/$name.pointwise.o larvanet_amd/csrc/wgrad3x3_mfma.o
echo tools/_diag/$name.so
