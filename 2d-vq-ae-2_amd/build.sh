#!/bin/bash
# Build libvqae_hip.so (gfx950 only) in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
mkdir -p build
objs=()
for f in csrc/vq_kernels.hip csrc/vq_filter.hip csrc/vq_proj.hip csrc/conv_mfma.hip csrc/conv_wino.hip csrc/conv_wino43.hip csrc/trunk16.hip csrc/misc_kernels.hip csrc/fixup_fused.hip csrc/down_fused.hip csrc/down16.hip csrc/up16.hip csrc/same8_16.hip csrc/stem16.hip csrc/mbconv.hip csrc/handle.hip; do
  o=build/$(basename "${f%.hip}").o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ csrc/common.h -nt "$o" ] || [ ../include/vqae_hip.h -nt "$o" ]; then
    "$HIPCC" --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wall -Wno-unused-function -c "$f" -o "$o" &
  fi
  objs+=("$o")
done
wait
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o libvqae_hip.so "${objs[@]}"
echo "built $(pwd)/libvqae_hip.so"
