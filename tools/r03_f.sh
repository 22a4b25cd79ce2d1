#!/bin/bash
set -u
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_blocks_gpu.py -k "wino16 or (per_block and f32) or (chains and f32)" tests/test_model_gpu.py -x -q > gpurun_out/r03/pytest_f.log 2>&1
rc=$?; echo "pytest rc $rc"; tail -5 gpurun_out/r03/pytest_f.log
[ $rc -eq 0 ] || exit 1
rm -rf gpurun_out/r03/prof_f
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r03/prof_f" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --no-other-configs > "$GRAFT_REPO_ROOT/gpurun_out/r03/f_bench.log" 2>&1) || { echo "bench failed"; tail -5 gpurun_out/r03/f_bench.log; exit 1; }
grep '^{' gpurun_out/r03/f_bench.log | tail -1 | cut -c1-400
f=$(ls gpurun_out/r03/prof_f/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
    print(f'  {float(r["Percentage"]):6.2f}% calls {r["Calls"]:>5} avg {float(r["AverageNs"])/1e3:8.1f}us  {n}')
PY
