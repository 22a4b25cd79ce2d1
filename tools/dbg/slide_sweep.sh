#!/bin/bash
mkdir -p gpurun_out/r03
for cfg in "100 14 2" "200 14 2" "100 14 4" "100 10 2" "256 14 2"; do
  set -- $cfg
  timeout -k 10 300 python tools/bench_slide.py --rows 250 --cols 400 --batch $1 --workers $2 --prefetch $3 --dtype f16 --loader ring > gpurun_out/r03/slide_sw_$1_$2_$3.log 2>&1 || { tail -5 gpurun_out/r03/slide_sw_$1_$2_$3.log; exit 1; }
  python3 - gpurun_out/r03/slide_sw_$1_$2_$3.log "$cfg" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[2], "->", d["patches_per_s"], "patches/s; encoder only", d["encoder_only_patches_per_s"], "; host", {k: round(v, 2) for k, v in d["stages"]["host_s"].items()}, "gpu", {k: round(v, 2) for k, v in d["stages"]["gpu_s"].items()}, flush=True)
PY
done
