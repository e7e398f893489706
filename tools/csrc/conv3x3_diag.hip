// Measurement build of the conv kernels: the PRODUCT's translation unit, unchanged, followed by the measurement entry
// points (kernel-attached launch timing, the pair-chain and layer-pipeline experiments of round 5).  tools/build_diag.sh
// compiles this file instead of larvanet_amd/csrc/conv3x3_mfma.hip; the product build never sees anything under tools/csrc.
#include "../../larvanet_amd/csrc/conv3x3_mfma.hip"
#include "conv3x3_diag_api.inc"
