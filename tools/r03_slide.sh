#!/bin/bash
# round 3: driver / parity tests touched this round + the whole-slide pipeline at BASELINE configs[4] size (100k tiles)
set -u
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_driver_gpu.py tests/test_configs_gpu.py tests/test_autocast_gpu.py "tests/test_blocks_gpu.py::test_production_blocks_match_oracle_per_block" -x -q > gpurun_out/r03/pytest_slide.log 2>&1
rc=$?; echo "pytest rc $rc"; tail -15 gpurun_out/r03/pytest_slide.log
[ $rc -eq 0 ] || exit 1
for cfg in "ring 14 3" "torch 12 5"; do
  set -- $cfg
  timeout -k 10 300 python tools/bench_slide.py --rows 64 --cols 128 --batch 100 --workers $2 --prefetch $3 --dtype f16 --loader $1 > gpurun_out/r03/slide_8k_$1.log 2>&1 || { echo "slide $1 failed"; tail -20 gpurun_out/r03/slide_8k_$1.log; exit 1; }
  tail -1 gpurun_out/r03/slide_8k_$1.log
done
timeout -k 10 400 python tools/bench_slide.py --rows 250 --cols 400 --batch 100 --workers 14 --prefetch 3 --dtype f16 --loader ring --out gpurun_out/r03/slide_100k.json > gpurun_out/r03/slide_100k.log 2>&1 || { echo "slide 100k failed"; tail -20 gpurun_out/r03/slide_100k.log; exit 1; }
tail -1 gpurun_out/r03/slide_100k.log
