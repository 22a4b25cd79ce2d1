#!/bin/bash
set -u
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_blocks_gpu.py -k "bf16 or f16" tests/test_autocast_gpu.py -x -q > gpurun_out/r03/pytest_h.log 2>&1
rc=$?; echo "pytest rc $rc"; tail -3 gpurun_out/r03/pytest_h.log
[ $rc -eq 0 ] || exit 1
for cfg in "A bf16 full" "B bf16 full"; do
  set -- $cfg
  rm -rf gpurun_out/r03/prof_h_$1
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r03/prof_h_$1" -- python3 "$GRAFT_REPO_ROOT/bench.py" --config $1 --dtype $2 --mode $3 --batch 256 --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > "$GRAFT_REPO_ROOT/gpurun_out/r03/h_$1.log" 2>&1) || { echo "rocprof $cfg failed"; exit 1; }
  grep '^{' gpurun_out/r03/h_$1.log | tail -1 | cut -c1-130
  f=$(ls gpurun_out/r03/prof_h_$1/*/*kernel_stats.csv | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("same_small16", "same8_16", "up16_kernel", "trunk16_kernel<128")):
        n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:44]
        print(f'  calls {r["Calls"]:>5} avg {float(r["AverageNs"])/1e3:8.1f}us max {float(r["MaxNs"])/1e3:8.1f}  {n}')
PY
done
