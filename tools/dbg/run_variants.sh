#!/bin/bash
# usage: tools/dbg/run_variants.sh name1 name2 ...   ('base' = the product library); fp32 headline bench per variant
set -u
mkdir -p gpurun_out/r03
for v in "$@"; do
  lib=2d-vq-ae-2_amd/build/var/libvqae_$v.so
  [ "$v" = base ] && lib=2d-vq-ae-2_amd/libvqae_hip.so
  VQAE_HIP_LIB=$PWD/$lib timeout -k 10 200 python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-other-configs ${BENCH_ARGS:-} > gpurun_out/r03/var_$v.log 2>&1 || { echo "bench failed $v"; tail -5 gpurun_out/r03/var_$v.log; exit 1; }
  python3 - gpurun_out/r03/var_$v.log $v <<'PY'
import json, sys
d = json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1])
print(f"{sys.argv[2]:>10}: {d['value']:9.1f} patches/s  {d['ms_per_step']:.2f} ms/step  dominant kernel avg {d['roofline']['avg_ms']*1e3:7.1f} us", flush=True)
PY
done
