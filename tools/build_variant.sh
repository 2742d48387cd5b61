#!/bin/bash
# Builds the device library with extra -D knobs into raytracer_amd/lib/variants/<name>.so (for A/B runs
# with RTHIP_LIBRARY=...).  Usage: bash tools/build_variant.sh <name> [-DKNOB=value ...]
set -eu
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p raytracer_amd/lib/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fPIC -shared \
  -Rpass-analysis=kernel-resource-usage "$@" -o raytracer_amd/lib/variants/$NAME.so raytracer_amd/csrc/rthip.hip 2> /tmp/variant_$NAME.log
grep -A8 "Function Name: _Z6k_pathILb0" /tmp/variant_$NAME.log | grep -E " VGPRs:|VGPRs Spill|ScratchSize|LDS Size" | sed 's/.*remark: *//' | tr '\n' ' '
echo " <- $NAME"
