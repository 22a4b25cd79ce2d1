#!/bin/bash
set -u
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
bash tools/r03_vq.sh || exit 1
timeout -k 10 600 python -m pytest tests/test_blocks_gpu.py -k "persistent_conv1" tests/test_driver_gpu.py -x -q > gpurun_out/r03/pytest_b.log 2>&1
rc=$?; echo "pytest rc $rc"; tail -3 gpurun_out/r03/pytest_b.log
[ $rc -eq 0 ] || exit 1
for v in "VQAE_CONV1_ONESHOT=0" "VQAE_CONV1_ONESHOT=1"; do
  env $v timeout -k 10 300 python bench.py --prof-class 2 --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > gpurun_out/r03/class2_$v.log 2>&1 || { echo "bench $v failed"; tail -5 gpurun_out/r03/class2_$v.log; exit 1; }
  echo "$v: $(tail -1 gpurun_out/r03/class2_$v.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["value"], r["avg_ms"], r["achieved"], r["frac"])')"
done
timeout -k 10 400 python tools/bench_slide.py --rows 250 --cols 400 --batch 100 --workers 14 --prefetch 2 --dtype f16 --loader ring --out gpurun_out/r03/slide_100k.json > gpurun_out/r03/slide_100k.log 2>&1 || { echo "slide 100k failed"; tail -20 gpurun_out/r03/slide_100k.log; exit 1; }
tail -1 gpurun_out/r03/slide_100k.log
