#!/usr/bin/env python3
"""Copy the judged summaries of the latest gpurun_out/ measurement set into profiles/ (round 1 naming):
PMC passes of tools/pmc_pass.sh (fetch / write / busy), the rocprofv3 --stats csv and the default bench log."""
import collections, csv, glob, json, os, shutil

def per_launch(name, match):
    f = max(glob.glob(f'gpurun_out/pmc_{name}/*/*_counter_collection.csv'), key=os.path.getmtime)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if match(r['Kernel_Name']):
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    out = {}
    for c, v in agg.items():
        big = [x for x in v if x >= 0.5 * max(v)]          # the B = 256 launches (calibration launches are smaller)
        out[c] = sum(big) / len(big)
    return out

m = lambda k: 'wino_trunk_kernel<128, 2, 0, false>' in k
f, w, b = per_launch('fetch', m), per_launch('write', m), per_launch('busy', m)
M = 256 * 32 * 32
fetch, write, alg = f['FETCH_SIZE'] * 1024 * 2, w['WRITE_SIZE'] * 1024, 4 * M * 128 * 4
d = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* in separate passes (tools/pmc_pass.sh) on `python bench.py --steps 1 "
             "--warmup 1 --no-cpu-baseline` (cfg B, B=256, fp32); FETCH_SIZE (KB) doubled (gfx950 reports 1/2 of wide coalesced "
             "reads, MI355X_MICROARCH.md HBM section); averages per launch of the trunk Fixup kernel "
             "wino_trunk_kernel<128, 2, fp32> (Winograd conv2 + fused conv3 + next conv1)",
     "class1_fetch_bytes_corrected": fetch, "class1_write_bytes": write, "class1_bytes_per_launch": fetch + write,
     "class1_algorithmic_bytes": alg,
     "class1_mfma_busy_frac": b['SQ_VALU_MFMA_BUSY_CYCLES'] / 4 / b['SQ_BUSY_CU_CYCLES'],
     "class1_insts_mfma_per_launch": b['SQ_INSTS_MFMA'], "class1_insts_valu_incl_mfma_per_launch": b['SQ_INSTS_VALU']}
json.dump(d, open('profiles/r01_pmc_traffic.json', 'w'), indent=1)
st = max(glob.glob('gpurun_out/prof_final/*/*_kernel_stats.csv'), key=os.path.getmtime)
shutil.copy(st, 'profiles/r01_bench_B256_kernel_stats_final.csv')
shutil.copy('gpurun_out/bench_final.log', 'profiles/r01_bench_final.log')
rows = list(csv.DictReader(open(st)))
out = {"source": "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-cpu-baseline (cfg B, B=256, fp32, 2 warm-up "
                 "+ 5 timed steps + the smaller calibration launches, which lower the per-kernel averages)",
       "kernels": [{"kernel": r["Name"][:110], "calls": int(r["Calls"]), "total_ms": round(int(r["TotalDurationNs"]) / 1e6, 3),
                    "avg_us": round(float(r["AverageNs"]) / 1e3, 2), "max_us": round(int(r["MaxNs"]) / 1e3, 1),
                    "pct": float(r["Percentage"])} for r in rows[:18]],
       "pmc_trunk_kernel": d}
json.dump(out, open('profiles/r01_summary.json', 'w'), indent=1)
print("traffic MB", round((fetch + write) / 1e6), "busy", round(d['class1_mfma_busy_frac'], 4))
for k in out["kernels"][:8]:
    print(k["pct"], k["calls"], k["avg_us"], k["max_us"], k["kernel"][:64])
l = [x for x in open('gpurun_out/bench_final.log') if x.startswith('{')][-1]
j = json.loads(l)
print(j['value'], j['roofline']['avg_ms'], j['roofline']['frac'], j['roofline']['traffic'])
