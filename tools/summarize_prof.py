#!/usr/bin/env python3
"""Summarise rocprofv3 output directories (kernel trace / PMC passes) into a small JSON for profiles/.

    python tools/summarize_prof.py --trace gpurun_out/prof --pmc gpurun_out/pmc_fetch gpurun_out/pmc_write ... -o profiles/x.json
"""
import argparse
import collections
import csv
import glob
import json


def short(name):
    if "conv_mfma_kernel" in name:
        return name[name.index("conv_mfma_kernel"):name.index(">") + 1]
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:60]


def cls(name, grid, dur_us):
    """Split the trunk conv launches (same kernel symbol) into 3x3 / 1x1 by duration."""
    s = short(name)
    if "conv_mfma_kernel<128, 32" in s and grid == "524288":
        return s + (" [trunk block]" if dur_us > 400 else " [trunk 1x1]")
    return s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trace")
    ap.add_argument("--pmc", nargs="*", default=[])
    ap.add_argument("-o", required=True)
    a = ap.parse_args()
    out = {}
    if a.trace:
        rows = list(csv.DictReader(open(glob.glob(a.trace + "/*/*_kernel_trace.csv")[0])))
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in rows:
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            k = cls(r["Kernel_Name"], r["Grid_Size_X"], d)
            agg[k][0] += 1
            agg[k][1] += d
        tot = sum(v[1] for v in agg.values())
        out["kernel_trace"] = {"total_ms": round(tot / 1e3, 3), "kernels": [
            {"kernel": k, "calls": v[0], "total_ms": round(v[1] / 1e3, 3), "avg_us": round(v[1] / v[0], 2),
             "pct": round(100 * v[1] / tot, 2)} for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]]}
    pm = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for d in a.pmc:
        for f in glob.glob(d + "/*/*_counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                k = cls(r["Kernel_Name"], r["Grid_Size"], dur)
                e = pm[k][r["Counter_Name"]]
                e[0] += 1
                e[1] += float(r["Counter_Value"])
    if pm:
        keep = sorted(pm.items(), key=lambda kv: -max(v[1] for v in kv[1].values()))[:12]
        out["pmc_avg_per_launch"] = {k: {c: round(v[1] / v[0], 2) for c, v in cs.items()} for k, cs in keep}
    json.dump(out, open(a.o, "w"), indent=1)
    print("wrote", a.o)


if __name__ == "__main__":
    main()
