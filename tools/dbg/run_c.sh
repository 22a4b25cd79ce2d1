#!/bin/bash
set -u
mkdir -p gpurun_out/r03
timeout -k 10 300 python -m pytest tests/test_model_gpu.py -k "full_configs" -x -q > gpurun_out/r03/pytest_c.log 2>&1; echo "pytest rc $?"; tail -2 gpurun_out/r03/pytest_c.log
for v in 1 0; do
  VQAE_WINO43=$v timeout -k 10 300 python3 bench.py --config C --dtype f32 --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs > gpurun_out/r03/c_f32_$v.log 2>&1 || { echo failed; tail -5 gpurun_out/r03/c_f32_$v.log; exit 1; }
  python3 - gpurun_out/r03/c_f32_$v.log $v <<'PY'
import json, sys
d = json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1])
print(f"cfg C f32 WINO43={sys.argv[2]}: {d['value']:9.1f} patches/s  {d['ms_per_step']:.2f} ms/step  trunk avg {d['roofline']['avg_ms']*1e3:7.1f} us", flush=True)
PY
done
