#!/usr/bin/env python3
"""Copy a round's measurement set from gpurun_out/<round> (scratch) into profiles/ (tracked): bench JSON lines, rocprofv3
--kernel-trace --stats summaries, PMC traffic / counter tables, the slide pipeline record, the parity report.

    python tools/finalize_round.py r03
"""
import glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
src, dst = os.path.join(ROOT, "gpurun_out", rnd), os.path.join(ROOT, "profiles")
lines = {}
for f in sorted(glob.glob(src + "/bench_*.log") + glob.glob(src + "/class*.log") + glob.glob(src + "/vq_proj_bench.log")):
    last = [l for l in open(f).read().splitlines() if l.startswith("{")]
    if last:
        lines[os.path.basename(f)[:-4]] = json.loads(last[-1])
json.dump(lines, open(os.path.join(dst, f"{rnd}_bench_lines.json"), "w"), indent=1)
for d in sorted(glob.glob(src + "/prof_*/")):
    tag = os.path.basename(d.rstrip("/"))[5:]
    stats = sorted(glob.glob(d + "*/*kernel_stats.csv"), key=os.path.getmtime)
    if stats:
        shutil.copy(stats[-1], os.path.join(dst, f"{rnd}_kernel_stats_{tag}.csv"))
for name in (f"{rnd}_pmc_traffic.json",):
    p = os.path.join(ROOT, "gpurun_out", name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, name))
pmc = {}
for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_{rnd}_*.json"))):
    pmc[os.path.basename(f)[4:-5]] = json.load(open(f))
if pmc:
    json.dump(pmc, open(os.path.join(dst, f"{rnd}_pmc_counters.json"), "w"), indent=1)
if os.path.exists(os.path.join(src, "slide_100k.json")):
    shutil.copy(os.path.join(src, "slide_100k.json"), os.path.join(dst, f"{rnd}_slide.json"))
if os.path.exists(os.path.join(ROOT, "gpurun_out", "parity_report.jsonl")):
    shutil.copy(os.path.join(ROOT, "gpurun_out", "parity_report.jsonl"), os.path.join(dst, f"{rnd}_parity_report.jsonl"))
for k, v in lines.items():
    r = v.get("roofline", {})
    if "value" in v:
        print(f"{k:28s} {v['value']:9.1f} patches/s  {v['ms_per_step']:8.2f} ms/step  {r.get('kernel', '')[:24]:24s} avg_ms {r.get('avg_ms')} "
              f"frac {r.get('frac')} hbm_frac {r.get('hbm_frac')} traffic {r.get('traffic')}")
    else:
        print(k, v)
