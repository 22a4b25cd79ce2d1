#!/bin/bash
# round 3, first GPU call: GPU test suite, the default bench line (with other_configs), slide bench baseline
set -u
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest_first.log 2>&1
echo "pytest rc $?"; tail -3 gpurun_out/r03/pytest_first.log
timeout -k 10 400 python bench.py > gpurun_out/r03/bench_default_first.log 2>&1 || { echo bench failed; tail -20 gpurun_out/r03/bench_default_first.log; exit 1; }
tail -1 gpurun_out/r03/bench_default_first.log | cut -c1-6000
timeout -k 10 300 python tools/bench_slide.py --rows 64 --cols 128 --batch 100 --workers 12 --dtype f16 > gpurun_out/r03/slide_first.log 2>&1 || { echo slide failed; tail -20 gpurun_out/r03/slide_first.log; exit 1; }
tail -1 gpurun_out/r03/slide_first.log
