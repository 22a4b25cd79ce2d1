#!/bin/bash
# usage: tools/pmc_run.sh <name> <kernel-substring> <counter> [<counter> ...] -- <python script and args>
# one rocprofv3 --pmc pass (kernel trace only, as gpurun requires); prints the per-launch average of every counter for
# the kernels whose name contains the substring.  Output under gpurun_out/pmc_<name>/.
name=$1; match=$2; shift 2
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
shift
export TMPDIR=/tmp
rm -rf "gpurun_out/pmc_$name"
(cd /tmp && timeout -k 10 400 rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_$name" -- python3 "$GRAFT_REPO_ROOT/$1" "${@:2}" > "$GRAFT_REPO_ROOT/gpurun_out/pmc_$name.log" 2>&1) || { echo "pmc pass $name failed"; tail -5 "gpurun_out/pmc_$name.log"; exit 1; }
python3 - "$name" "$match" <<'PY'
import csv, glob, collections, sys, json
name, match = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(f"gpurun_out/pmc_{name}/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if match not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
        e = agg[k][r["Counter_Name"]]; e[0] += 1; e[1] += float(r["Counter_Value"])
res = {k: dict({c: round(v[1] / v[0], 1) for c, v in cs.items()}, launches=max(v[0] for v in cs.values())) for k, cs in agg.items()}
print(json.dumps(res))
json.dump(res, open(f"gpurun_out/pmc_{name}.json", "w"), indent=1)
PY
