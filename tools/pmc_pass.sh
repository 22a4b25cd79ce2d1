#!/bin/bash
# usage: tools/pmc_pass.sh <name> <counter> [<counter> ...]   (one rocprofv3 --pmc pass over a short bench run)
name=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_$name
timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc_$name -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_$name.log 2>&1
ls gpurun_out/pmc_$name/*/ | head -5
