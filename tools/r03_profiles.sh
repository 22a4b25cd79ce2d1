#!/bin/bash
# Round-3 measurement set (run on the GPU box from the repo root): PMC traffic, bench lines, rocprofv3 kernel stats, the whole-slide
# pipeline, the projected-VQ micro-benchmark with its counters.  Results land in gpurun_out/r03/; tools/finalize_profiles.py r03
# copies the summaries into profiles/.
set -u
mkdir -p gpurun_out/r03
export TMPDIR=/tmp VQAE_ROUND=r03
b() { # tag config dtype mode extra...
  local t=$1 c=$2 d=$3 m=$4; shift 4
  timeout -k 10 400 python bench.py --config $c --dtype $d --batch 256 --mode $m --steps 10 --warmup 3 --no-other-configs "$@" > gpurun_out/r03/bench_$t.log 2>&1
  local rc=$?
  tail -1 gpurun_out/r03/bench_$t.log | cut -c1-220
  [ $rc -eq 0 ] || { echo "bench $t failed (rc $rc): no further GPU step"; exit 1; }
}
# PMC traffic first: the bench lines below read profiles/r03_pmc_traffic.json (keyed by the kernel source hash)
for cfg in "B f32" "B bf16" "A bf16" "C f16"; do
  set -- $cfg
  timeout -k 10 700 python tools/pmc_traffic.py --config $1 --dtype $2 --batch 256 | cut -c1-260 || exit 1
done
b B_f32_full B f32 full
b B_bf16_full B bf16 full --no-cpu-baseline
b B_f16_full B f16 full --no-cpu-baseline
b A_bf16_full A bf16 full --no-cpu-baseline
b C_f16_full C f16 full --no-cpu-baseline
b B_f32_encode B f32 encode --no-cpu-baseline
b B_f16_encode B f16 encode --no-cpu-baseline
b A_f16_encode A f16 encode --no-cpu-baseline
b C_f16_encode C f16 encode --no-cpu-baseline
b class2_conv1 B f32 full --no-cpu-baseline --prof-class 2
b class3_vq_B B f32 full --no-cpu-baseline --prof-class 3
b class3_vq_A A bf16 full --no-cpu-baseline --prof-class 3
for cfg in "B f32 full" "B bf16 full" "A bf16 full" "C f16 full" "A f16 encode"; do
  set -- $cfg
  rm -rf gpurun_out/r03/prof_$1_$2_$3
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r03/prof_$1_$2_$3" -- python3 "$GRAFT_REPO_ROOT/bench.py" --config $1 --dtype $2 --mode $3 --batch 256 --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > "$GRAFT_REPO_ROOT/gpurun_out/r03/prof_$1_$2_$3.log" 2>&1) || { echo "rocprof $cfg failed"; exit 1; }
done
timeout -k 10 300 python tools/vq_proj_bench.py > gpurun_out/r03/vq_proj_bench.log 2>&1 || exit 1
tail -1 gpurun_out/r03/vq_proj_bench.log
bash tools/pmc_run.sh r03_vq16_a vq_proj16 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS -- tools/vq_proj_bench.py --reps 3 > gpurun_out/r03/pmc_vq16_a.json || exit 1
bash tools/pmc_run.sh r03_vq16_b vq_proj16 SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES -- tools/vq_proj_bench.py --reps 3 > gpurun_out/r03/pmc_vq16_b.json || exit 1
bash tools/pmc_run.sh r03_vq16_c vq_proj16 GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum -- tools/vq_proj_bench.py --reps 3 > gpurun_out/r03/pmc_vq16_c.json || echo "(TCC/TCP pass not available)"
timeout -k 10 400 python tools/bench_slide.py --rows 250 --cols 400 --batch 100 --workers 8 --prefetch 2 --dtype f16 --loader ring --out gpurun_out/r03/slide_100k.json > gpurun_out/r03/slide_100k.log 2>&1 || { echo "slide failed"; exit 1; }
tail -1 gpurun_out/r03/slide_100k.log | cut -c1-400
# the default line (what the driver runs), last: with the traffic figures in place
timeout -k 10 400 python bench.py > gpurun_out/r03/bench_default.log 2>&1; tail -1 gpurun_out/r03/bench_default.log | cut -c1-600
