#!/bin/bash
# usage: tools/dbg/build_variant.sh NAME file.hip "-Dflags"  -> 2d-vq-ae-2_amd/build/var/libvqae_NAME.so (the rest from build/*.o)
set -e
cd /root/repo/2d-vq-ae-2_amd
mkdir -p build/var
base=$(basename "${2%.hip}")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wall -Wno-unused-function $3 -c "csrc/$2" -o "build/var/${base}_$1.o"
objs=()
for n in vq_kernels vq_filter vq_proj conv_mfma conv_wino conv_wino43 trunk16 misc_kernels fixup_fused down_fused down16 up16 same8_16 stem16 mbconv handle; do
  o=build/$n.o
  [ "$(basename $o)" = "$base.o" ] && objs+=("build/var/${base}_$1.o") || objs+=("$o")
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "build/var/libvqae_$1.so" "${objs[@]}"
echo "built build/var/libvqae_$1.so"
