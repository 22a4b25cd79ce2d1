#!/bin/bash
set -u
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_vq_gpu.py -x -q > gpurun_out/r03/pytest_vq.log 2>&1
rc=$?; echo "pytest rc $rc"; tail -3 gpurun_out/r03/pytest_vq.log
[ $rc -eq 0 ] || exit 1
for v in "VQAE_VQ16_WPS=4" "VQAE_VQ16_WPS=3" "VQAE_VQ16_WPS=4 VQAE_VQ16_NOFILTER=1"; do
  echo "$v: $(env $v timeout -k 10 300 python tools/vq_proj_bench.py | cut -c1-200)"
done
VQAE_VQ16_WPS=3 timeout -k 10 300 python -m pytest tests/test_vq_gpu.py -x -q -k projected 2>&1 | tail -2
bash tools/pmc_run.sh vq16f_a vq_proj16 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS -- tools/vq_proj_bench.py --reps 3 || exit 1
