#!/bin/bash
set -u
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_conv_gpu.py tests/test_model_gpu.py tests/test_configs_gpu.py -x -q > gpurun_out/r03/pytest_g.log 2>&1
rc=$?; echo "pytest rc $rc"; tail -5 gpurun_out/r03/pytest_g.log
[ $rc -eq 0 ] || exit 1
rm -rf gpurun_out/r03/prof_g
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r03/prof_g" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --no-other-configs > "$GRAFT_REPO_ROOT/gpurun_out/r03/g_bench.log" 2>&1) || { echo "bench failed"; tail -5 gpurun_out/r03/g_bench.log; exit 1; }
grep '^{' gpurun_out/r03/g_bench.log | tail -1 | cut -c1-330
f=$(ls gpurun_out/r03/prof_g/*/*kernel_stats.csv | head -1)
grep -E "stem|conv3x3_direct|up_tail|wino_trunk_kernel<128, 2" $f | cut -c1-70,180-300
