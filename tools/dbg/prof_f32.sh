#!/bin/bash
# rocprofv3 kernel stats of the fp32 headline; prints the top kernels
set -u
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
rm -rf gpurun_out/r03/prof_dbg
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r03/prof_dbg" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs ${BENCH_ARGS:-} > "$GRAFT_REPO_ROOT/gpurun_out/r03/prof_dbg.log" 2>&1) || { echo "rocprof failed"; tail -5 gpurun_out/r03/prof_dbg.log; exit 1; }
grep '^{' gpurun_out/r03/prof_dbg.log | tail -1 | cut -c1-130
f=$(ls gpurun_out/r03/prof_dbg/*/*kernel_stats.csv | head -1)
cp "$f" gpurun_out/r03/prof_dbg_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
    print(f'  calls {r["Calls"]:>5} avg {float(r["AverageNs"])/1e3:8.1f}us tot {float(r["TotalDurationNs"])/1e6:8.2f}ms {r["Percentage"]:>6}%  {n}')
PY
