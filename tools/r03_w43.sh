#!/bin/bash
set -u
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_model_gpu.py -k "wino43 or winograd_trunk or full_configs or native_forward" -x -q -s > gpurun_out/r03/pytest_w43.log 2>&1
rc=$?; echo "pytest rc $rc"; grep -E "F\(4,3\)|passed|failed|Error|error" gpurun_out/r03/pytest_w43.log | tail -15
[ $rc -eq 0 ] || exit 1
for v in 1 0; do
  VQAE_WINO43=$v timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > gpurun_out/r03/w43_bench_$v.log 2>&1 || { echo "bench failed"; tail -5 gpurun_out/r03/w43_bench_$v.log; exit 1; }
  python3 - gpurun_out/r03/w43_bench_$v.log $v <<'PY'
import json, sys
d = json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1])
print(f"WINO43={sys.argv[2]}: {d['value']:9.1f} patches/s  {d['ms_per_step']:.2f} ms/step  trunk avg {d['roofline']['avg_ms']*1e3:7.1f} us  idx_agreement {d.get('parity_on_cpu_sample')}", flush=True)
PY
done
