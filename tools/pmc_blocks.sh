#!/bin/bash
# usage: tools/pmc_blocks.sh <name> "<bench_blocks args>" <counter> [<counter> ...]
# one rocprofv3 --pmc pass over the block micro-benchmark (tools/bench_blocks.py); output under gpurun_out/pmc_<name>
name=$1; args=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_$name
timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc_$name -- python tools/bench_blocks.py $args --reps 1 > gpurun_out/pmc_$name.log 2>&1
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("gpurun_out/pmc_$name/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
        e = agg[k][r["Counter_Name"]]; e[0] += 1; e[1] += float(r["Counter_Value"])
for k, cs in sorted(agg.items(), key=lambda kv: -sum(v[0] for v in kv[1].values()))[:4]:
    print(k, {c: round(v[1] / v[0], 1) for c, v in cs.items()}, "launches", max(v[0] for v in cs.values()))
PY
