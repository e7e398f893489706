#!/bin/bash
# The measurement library: liblarva_hip.so's sources + the measurement entry points under tools/csrc/ (declared in
# tools/larva_diag.h: kernel-attached launch timing, the stamp / delay / clock-probe marker launches, the pair-chain and
# layer-pipeline experiments), into tools/_diag/<name>.so.  tools/csrc/conv3x3_diag.hip = the product's conv translation
# unit + conv3x3_diag_api.inc; larva_pointwise.hip is recompiled as it is (extra -D options go to both);
# tools/csrc/larva_markers.hip stands alone; the weight-gradient object is the product's (python -m larvanet_amd.build first).
#   tools/build_diag.sh diag [-DLARVA_DIAG_ONLY48=1 ...]
#   LARVA_HIP_LIB=tools/_diag/diag.so python tools/probe_pair_chain.py
set -euo pipefail
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/_diag
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Ilarvanet_amd/csrc -Itools/csrc"
hipcc $FLAGS "$@" -c tools/csrc/conv3x3_diag.hip -o tools/_diag/$name.conv.o &
hipcc $FLAGS "$@" -c larvanet_amd/csrc/larva_pointwise.hip -o tools/_diag/$name.pointwise.o &
hipcc $FLAGS -c tools/csrc/larva_markers.hip -o tools/_diag/$name.markers.o &
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_diag/$name.so tools/_diag/$name.conv.o tools/_diag/$name.pointwise.o tools/_diag/$name.markers.o larvanet_amd/csrc/wgrad3x3_mfma.o
echo tools/_diag/$name.so
