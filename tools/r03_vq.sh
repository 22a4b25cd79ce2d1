#!/bin/bash
# round 3: fused projected quantiser (vq_proj16_kernel) -- parity tests, then the class-3 roofline line for each variant
set -u
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_vq_gpu.py -x -q > gpurun_out/r03/pytest_vq.log 2>&1
rc=$?; echo "pytest rc $rc"; tail -5 gpurun_out/r03/pytest_vq.log
[ $rc -eq 0 ] || exit 1
for v in "VQAE_VQ16_WPS=3" "VQAE_VQ16_WPS=4" "VQAE_VQ_PROJ_V1=1"; do
  env $v timeout -k 10 300 python bench.py --config A --dtype bf16 --prof-class 3 --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > gpurun_out/r03/class3_$v.log 2>&1 || { echo "bench $v failed"; tail -5 gpurun_out/r03/class3_$v.log; exit 1; }
  echo "$v: $(tail -1 gpurun_out/r03/class3_$v.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["value"], r["avg_ms"], r["achieved"], r["frac"])')"
done
