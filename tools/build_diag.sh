#!/bin/bash
# The measurement library: liblarva_hip.so's sources + the entry points that exist only under -DLARVA_DIAG_API
# (csrc/*.inc), into tools/_diag/<name>.so.  Only conv3x3_mfma.hip is recompiled (the other two objects are the
# product's: python -m larvanet_amd.build first); extra -D options go to that compile.
#   tools/build_diag.sh diag [-DLARVA_DIAG_ONLY48=1 ...]
#   LARVA_HIP_LIB=tools/_diag/diag.so python tools/probe_pair_chain.py
set -euo pipefail
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/_diag
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DLARVA_DIAG_API=1 "$@" -Ilarvanet_amd/csrc -c larvanet_amd/csrc/conv3x3_mfma.hip -o tools/_diag/$name.conv.o
hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_diag/$name.so tools/_diag/$name.conv.o larvanet_amd/csrc/wgrad3x3_mfma.o larvanet_amd/csrc/larva_pointwise.o
echo tools/_diag/$name.so
