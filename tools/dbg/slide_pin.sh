#!/bin/bash
mkdir -p gpurun_out/r03
for cfg in "100 10 2" "100 8 2"; do
  set -- $cfg
  timeout -k 10 300 python tools/bench_slide.py --rows 250 --cols 400 --batch $1 --workers $2 --prefetch $3 --dtype f16 --loader ring > gpurun_out/r03/slide_one.log 2>&1 || { tail -5 gpurun_out/r03/slide_one.log; exit 1; }
  grep -c "Exception in thread\|could not be page-locked" gpurun_out/r03/slide_one.log
  python3 - gpurun_out/r03/slide_one.log "$cfg" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[2], "->", d["patches_per_s"], "patches/s in", d["seconds"], "s; encoder only", d["encoder_only_patches_per_s"], "; host", {k: round(v, 2) for k, v in d["stages"]["host_s"].items()}, "gpu", {k: round(v, 2) for k, v in d["stages"]["gpu_s"].items()}, flush=True)
PY
done
timeout -k 10 400 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --other-steps 1 > gpurun_out/r03/bd3.log 2> gpurun_out/r03/bd3.err
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r03/bd3.log") if l.startswith("{")][-1])
s = d["other_configs"].get("configs[4]_slide_pipeline_1gpu")
print("in bench:", s.get("value"), s.get("seconds"), s.get("error"), s.get("stages"))
PY
