#!/bin/bash
mkdir -p gpurun_out/r03
t0=$(date +%s.%N)
timeout -k 10 400 python bench.py > gpurun_out/r03/bench_default.log 2> gpurun_out/r03/bench_default.err; echo "rc $?"
t1=$(date +%s.%N); echo "wall $(echo "$t1 - $t0" | bc) s"
grep -E "leg" gpurun_out/r03/bench_default.err | tail -12
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r03/bench_default.log") if l.startswith("{")][-1])
print(d["value"], {k: (v.get("value") or v.get("error")) for k, v in d["other_configs"].items()})
print(json.dumps(d["other_configs"].get("configs[4]_slide_pipeline_1gpu"))[:900])
PY
