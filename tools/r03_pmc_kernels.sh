#!/bin/bash
# round 3: issue-side counters of the dominant kernels (two passes per configuration; kernel trace only, as gpurun requires)
set -u
export TMPDIR=/tmp
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"
B="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VMEM GRBM_GUI_ACTIVE"
bash tools/pmc_run.sh r03_f32_a _kernel $A -- bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs > /dev/null || exit 1
bash tools/pmc_run.sh r03_f32_b _kernel $B -- bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs > /dev/null || exit 1
bash tools/pmc_run.sh r03_bf16_a _kernel $A -- bench.py --dtype bf16 --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs > /dev/null || exit 1
bash tools/pmc_run.sh r03_bf16_b _kernel $B -- bench.py --dtype bf16 --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs > /dev/null || exit 1
ls -la gpurun_out/pmc_r03_*.json
