#!/bin/bash
mkdir -p gpurun_out/r03
for only in ""; do
  VQAE_BENCH_ONLY_SLIDE=$only timeout -k 10 400 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --other-steps 1 > gpurun_out/r03/bd3.log 2> gpurun_out/r03/bd3.err
  grep "slide_pipeline" gpurun_out/r03/bd3.err | tail -1
  python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r03/bd3.log") if l.startswith("{")][-1])
s = d["other_configs"].get("configs[4]_slide_pipeline_1gpu")
print("in bench:", s.get("value"), s.get("seconds"), s.get("error"), {k: round(v, 2) for k, v in s.get("stages", {}).get("host_s", {}).items()})
PY
done
