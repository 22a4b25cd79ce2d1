#!/usr/bin/env python3
"""Copy the round-2 measurement set from gpurun_out/r02 (scratch) into profiles/ (tracked): bench JSON lines,
rocprofv3 --kernel-trace --stats summaries (top kernels), PMC traffic."""
import csv, glob, json, os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, "gpurun_out", "r02"), os.path.join(ROOT, "profiles")
lines = {}
for f in sorted(glob.glob(src + "/bench_*.log") + glob.glob(src + "/class*.log")):
    last = [l for l in open(f).read().splitlines() if l.startswith("{")]
    if last:
        lines[os.path.basename(f)[:-4]] = json.loads(last[-1])
json.dump(lines, open(os.path.join(dst, "r02_bench_lines.json"), "w"), indent=1)
for d in sorted(glob.glob(src + "/prof_*/")):
    tag = os.path.basename(d.rstrip("/"))[5:]
    stats = sorted(glob.glob(d + "*/*kernel_stats.csv"), key=os.path.getmtime)
    if stats:
        shutil.copy(stats[-1], os.path.join(dst, f"r02_kernel_stats_{tag}.csv"))
if os.path.exists(os.path.join(ROOT, "gpurun_out", "r02_pmc_traffic.json")):
    shutil.copy(os.path.join(ROOT, "gpurun_out", "r02_pmc_traffic.json"), os.path.join(dst, "r02_pmc_traffic.json"))
if os.path.exists(os.path.join(ROOT, "gpurun_out", "parity_report.jsonl")):
    shutil.copy(os.path.join(ROOT, "gpurun_out", "parity_report.jsonl"), os.path.join(dst, "r02_parity_report.jsonl"))
for k, v in lines.items():
    r = v.get("roofline", {})
    print(f"{k:28s} {v['value']:9.1f} patches/s  {v['ms_per_step']:8.2f} ms/step  roofline {r.get('achieved')} {r.get('unit')} frac {r.get('frac')}")
