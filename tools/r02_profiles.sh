#!/bin/bash
# Round-2 measurement set (run on the GPU box from the repo root): bench lines, rocprofv3 kernel stats, PMC traffic.
set -u
mkdir -p gpurun_out/r02
export TMPDIR=/tmp
for cfg in "B f32 256" "B bf16 256" "A bf16 256" "C f16 256"; do
  set -- $cfg
  extra=""; [ "$2" = "f32" ] || extra="--no-cpu-baseline"
  timeout -k 10 400 python bench.py --config $1 --dtype $2 --batch $3 --steps 10 --warmup 3 $extra > gpurun_out/r02/bench_$1_$2.log 2>&1
  tail -1 gpurun_out/r02/bench_$1_$2.log | cut -c1-400
done
for cfg in "B f32" "B bf16" "C f16"; do
  set -- $cfg
  rm -rf gpurun_out/r02/prof_$1_$2
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r02/prof_$1_$2" -- python "$GRAFT_REPO_ROOT/bench.py" --config $1 --dtype $2 --batch 256 --steps 5 --warmup 2 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/r02/prof_$1_$2.log" 2>&1)
  f=$(ls gpurun_out/r02/prof_$1_$2/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && head -8 "$f" | cut -c1-200
done
timeout -k 10 700 python tools/pmc_traffic.py --config B --dtype f32 --batch 256 | cut -c1-400
timeout -k 10 700 python tools/pmc_traffic.py --config B --dtype bf16 --batch 256 | cut -c1-400
