#!/bin/bash
# usage: tools/dbg/run_variants16.sh name1 name2 ...  ('base' = the product library): cfg A f16 encode-only and cfg B bf16 full per variant
set -u
mkdir -p gpurun_out/r03
for v in "$@"; do
  lib=2d-vq-ae-2_amd/build/var/libvqae_$v.so
  [ "$v" = base ] && lib=2d-vq-ae-2_amd/libvqae_hip.so
  for cfg in "A f16 encode" "B bf16 full"; do
    set -- $cfg
    VQAE_HIP_LIB=$PWD/$lib timeout -k 10 200 python3 bench.py --config $1 --dtype $2 --mode $3 --steps 8 --warmup 3 --no-cpu-baseline --no-other-configs > gpurun_out/r03/var16_$v.log 2>&1 || { echo "bench failed $v"; tail -5 gpurun_out/r03/var16_$v.log; exit 1; }
    python3 - gpurun_out/r03/var16_$v.log $v "$cfg" <<'PY'
import json, sys
d = json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1])
print(f"{sys.argv[2]:>10} {sys.argv[3]:>14}: {d['value']:9.1f} patches/s  {d['ms_per_step']:.2f} ms/step", flush=True)
PY
  done
done
