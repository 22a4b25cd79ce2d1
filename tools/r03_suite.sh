#!/bin/bash
# full GPU suite + the default bench line
set -u
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest_suite.log 2>&1
rc=$?; echo "pytest rc $rc"; tail -3 gpurun_out/r03/pytest_suite.log
[ $rc -eq 0 ] || { grep -E "^(FAILED|ERROR)|Error|assert" gpurun_out/r03/pytest_suite.log | head -20; exit 1; }
timeout -k 10 400 python bench.py > gpurun_out/r03/bench_default_suite.log 2>&1 || { echo bench failed; tail -20 gpurun_out/r03/bench_default_suite.log; exit 1; }
tail -1 gpurun_out/r03/bench_default_suite.log | cut -c1-1500
