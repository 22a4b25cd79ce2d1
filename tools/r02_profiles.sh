#!/bin/bash
# Round-2 measurement set (run on the GPU box from the repo root): bench lines, rocprofv3 kernel stats, PMC traffic.
# Results land in gpurun_out/r02/; tools/finalize_r02.py copies the summaries into profiles/.
set -u
mkdir -p gpurun_out/r02
export TMPDIR=/tmp
b() { # config dtype batch mode extra...
  local c=$1 d=$2 n=$3 m=$4; shift 4
  timeout -k 10 400 python bench.py --config $c --dtype $d --batch $n --mode $m --steps 10 --warmup 3 "$@" > gpurun_out/r02/bench_${c}_${d}_${m}.log 2>&1
  local rc=$?
  tail -1 gpurun_out/r02/bench_${c}_${d}_${m}.log | cut -c1-330
  [ $rc -eq 0 ] || { echo "bench $c $d $m failed (rc $rc): no further GPU step"; exit 1; }
}
# PMC traffic first: the bench lines below read profiles/r02_pmc_traffic.json (keyed by the kernel source hash)
timeout -k 10 700 python tools/pmc_traffic.py --config B --dtype f32 --batch 256 | cut -c1-300 || exit 1
timeout -k 10 700 python tools/pmc_traffic.py --config B --dtype bf16 --batch 256 | cut -c1-300 || exit 1
b B f32 256 full
b B bf16 256 full --no-cpu-baseline
b B f16 256 full --no-cpu-baseline
b A bf16 256 full --no-cpu-baseline
b A f32 64 full --no-cpu-baseline
b C f16 256 full --no-cpu-baseline
b B f32 256 encode --no-cpu-baseline
b B f16 256 encode --no-cpu-baseline
b A f16 256 encode --no-cpu-baseline
b C f16 256 encode --no-cpu-baseline
timeout -k 10 300 python bench.py --prof-class 2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02/class2_conv1.log 2>&1 || exit 1
timeout -k 10 300 python bench.py --prof-class 3 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02/class3_vq_B.log 2>&1 || exit 1
timeout -k 10 300 python bench.py --config A --dtype bf16 --prof-class 3 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02/class3_vq_A.log 2>&1 || exit 1
for cfg in "B f32" "B bf16" "A bf16" "C f16"; do
  set -- $cfg
  rm -rf gpurun_out/r02/prof_$1_$2
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r02/prof_$1_$2" -- python "$GRAFT_REPO_ROOT/bench.py" --config $1 --dtype $2 --batch 256 --steps 5 --warmup 2 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/r02/prof_$1_$2.log" 2>&1) || exit 1
done
# the default line (what the driver runs)
timeout -k 10 400 python bench.py > gpurun_out/r02/bench_default.log 2>&1; tail -1 gpurun_out/r02/bench_default.log | cut -c1-2500
