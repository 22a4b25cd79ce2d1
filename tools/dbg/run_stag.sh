#!/bin/bash
set -u
mkdir -p gpurun_out/r03
for st in 0 3 6 9 12 18 0; do
  VQAE_W43_STAG=$st VQAE_HIP_LIB=$PWD/2d-vq-ae-2_amd/build/var/libvqae_e1s.so timeout -k 10 200 python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-other-configs > gpurun_out/r03/stag_$st.log 2>&1 || { echo failed; tail -3 gpurun_out/r03/stag_$st.log; exit 1; }
  python3 - gpurun_out/r03/stag_$st.log $st <<'PY'
import json, sys
d = json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1])
print(f"stag {sys.argv[2]:>3}: {d['value']:9.1f} patches/s  dominant kernel avg {d['roofline']['avg_ms']*1e3:7.1f} us", flush=True)
PY
done
