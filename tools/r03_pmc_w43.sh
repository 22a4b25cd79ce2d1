#!/bin/bash
# issue-side counters + HBM traffic of the F(4x4,3x3) trunk kernel (separate passes; kernel trace only, as gpurun requires)
set -u
export TMPDIR=/tmp
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"
B="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VMEM GRBM_GUI_ACTIVE"
bash tools/pmc_run.sh r03w_f32_a wino43_trunk $A -- bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs || exit 1
bash tools/pmc_run.sh r03w_f32_b wino43_trunk $B -- bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs || exit 1
python3 tools/pmc_traffic.py --config B --dtype f32 --batch 256 || exit 1
