#!/bin/bash
mkdir -p gpurun_out/r03
for extra in "--no-cpu-baseline" ""; do
  timeout -k 10 400 python bench.py $extra --steps 3 --warmup 1 --other-steps 2 > gpurun_out/r03/bd2.log 2> gpurun_out/r03/bd2.err; echo "rc $? [$extra]"
  python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r03/bd2.log") if l.startswith("{")][-1])
s = d["other_configs"].get("configs[4]_slide_pipeline_1gpu")
print(d["value"], s.get("value"), s.get("seconds"), s.get("error"))
PY
done
timeout -k 10 300 python tools/bench_slide.py --rows 100 --cols 200 --batch 100 --workers 8 --prefetch 2 --dtype f16 --loader ring | tail -1 | cut -c1-300
