#!/bin/bash
set -u
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
bash tools/pmc_run.sh vq16_a vq_proj SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS -- tools/vq_proj_bench.py --reps 3 || exit 1
bash tools/pmc_run.sh vq16_b vq_proj SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES -- tools/vq_proj_bench.py --reps 3 || exit 1
bash tools/pmc_run.sh vq16_c vq_proj GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES -- tools/vq_proj_bench.py --reps 3 || exit 1
export VQAE_VQ_PROJ_V1=1
bash tools/pmc_run.sh vq1_a vq_proj SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS -- tools/vq_proj_bench.py --reps 3 || exit 1
unset VQAE_VQ_PROJ_V1
timeout -k 10 600 python -m pytest tests/test_driver_gpu.py -x -q 2>&1 | tail -3
