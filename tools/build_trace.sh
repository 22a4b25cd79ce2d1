#!/bin/bash
# Developer build: libvqae_hip_trace.so = the normal objects + trunk16.hip with per-phase s_memtime stamps (-DVQAE_T16_TRACE).
set -euo pipefail
cd "$(dirname "$0")/../2d-vq-ae-2_amd"
bash build.sh >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -DVQAE_T16_TRACE -c csrc/trunk16.hip -o build/trunk16_trace.o
objs=$(ls build/*.o | grep -v trunk16)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libvqae_hip_trace.so $objs build/trunk16_trace.o
echo "built $(pwd)/libvqae_hip_trace.so"
