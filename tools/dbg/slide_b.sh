#!/bin/bash
mkdir -p gpurun_out/r03
for cfg in "100 10 2" "128 10 2" "200 10 2" "256 10 2" "256 8 1"; do
  set -- $cfg
  timeout -k 10 300 python tools/bench_slide.py --rows 250 --cols 400 --batch $1 --workers $2 --prefetch $3 --dtype f16 --loader ring > gpurun_out/r03/slide_one.log 2>&1 || { tail -5 gpurun_out/r03/slide_one.log; exit 1; }
  python3 - gpurun_out/r03/slide_one.log "$cfg" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[2], "->", d["patches_per_s"], "patches/s in", d["seconds"], "s; encoder only", d["encoder_only_patches_per_s"], "; gpu", {k: round(v, 2) for k, v in d["stages"]["gpu_s"].items()}, "wait", round(d["stages"]["host_s"]["loader_wait"], 2), flush=True)
PY
done
