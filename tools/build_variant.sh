#!/bin/bash
# Build liblarva_hip.so with extra -D options into tools/_diag/<name>.so for same-box A/B timing:
#   tools/build_variant.sh nodiag -DLARVA_DIAG=16
#   LARVA_HIP_LIB=tools/_diag/nodiag.so python bench.py ...
set -euo pipefail
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/_diag
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$@" -Ilarvanet_amd/csrc \
  larvanet_amd/csrc/conv3x3_mfma.hip larvanet_amd/csrc/wgrad3x3_mfma.hip \
  larvanet_amd/csrc/larva_pointwise.hip -o tools/_diag/$name.so
echo tools/_diag/$name.so
