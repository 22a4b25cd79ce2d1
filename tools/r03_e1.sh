#!/bin/bash
# E1: half-size Winograd workgroups (2 waves, 4 per CU) at C = 32 / 64 -- libvqae_hip_exp.so vs the default build
set -u
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
for lib in "" "$GRAFT_REPO_ROOT/2d-vq-ae-2_amd/libvqae_hip_exp.so"; do
  tag=$([ -z "$lib" ] && echo base || echo smallwg)
  rm -rf gpurun_out/r03/prof_e1_$tag
  (cd /tmp && VQAE_HIP_LIB=$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r03/prof_e1_$tag" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > "$GRAFT_REPO_ROOT/gpurun_out/r03/e1_$tag.log" 2>&1) || { echo "e1 $tag failed"; tail -5 gpurun_out/r03/e1_$tag.log; exit 1; }
  echo "$tag: $(tail -1 gpurun_out/r03/e1_$tag.log | cut -c1-200)"
  f=$(ls gpurun_out/r03/prof_e1_$tag/*/*kernel_stats.csv | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
    print(f'  {float(r["Percentage"]):6.2f}% calls {r["Calls"]:>5} avg {float(r["AverageNs"])/1e3:8.1f}us  {n}')
PY
done
VQAE_HIP_LIB=$GRAFT_REPO_ROOT/2d-vq-ae-2_amd/libvqae_hip_exp.so timeout -k 10 900 python -m pytest tests/test_blocks_gpu.py -k "f32" tests/test_model_gpu.py -x -q 2>&1 | tail -3
