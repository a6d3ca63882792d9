#!/bin/bash
# Build a variant of libmimo_hip.so with extra flags for ONE source file (A/B and timing-only ablation builds):
#   bash scripts/build_variant.sh <name> <source stem> <flags...>   ->  build/variants/libmimo_<name>.so
set -eu
NAME=$1; STEM=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$R/build/variants"
O=$R/build/variants/${STEM}_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function "$@" -c "$R/mimo_unet_amd/csrc/$STEM.hip" -o "$O"
OBJS=""
for f in conv3x3 conv_thin conv_bf16x3 conv_wide wgrad_split elementwise optim plan ops_api; do
  if [ "$f" = "$STEM" ]; then OBJS="$OBJS $O"; else OBJS="$OBJS $R/mimo_unet_amd/csrc/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script="$R/mimo_unet_amd/csrc/exports.map" -o "$R/build/variants/libmimo_$NAME.so" $OBJS
echo "$R/build/variants/libmimo_$NAME.so"
