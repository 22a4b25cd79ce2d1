#!/bin/bash
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_driver_gpu.py tests/test_configs_gpu.py -x -q > gpurun_out/r03/pytest_driver.log 2>&1; echo "pytest rc $?"; tail -2 gpurun_out/r03/pytest_driver.log
for eb in auto none auto none; do
  timeout -k 10 300 python tools/bench_slide.py --rows 250 --cols 400 --batch 100 --workers 10 --prefetch 2 --dtype f16 --loader ring --encode-batch $eb > gpurun_out/r03/slide_one.log 2>&1 || { tail -5 gpurun_out/r03/slide_one.log; exit 1; }
  python3 - gpurun_out/r03/slide_one.log "$eb" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("encode_batch", sys.argv[2], "->", d["patches_per_s"], "patches/s in", d["seconds"], "s; encoder only", d["encoder_only_patches_per_s"], "; gpu", {k: round(v, 2) for k, v in d["stages"]["gpu_s"].items()}, flush=True)
PY
done
