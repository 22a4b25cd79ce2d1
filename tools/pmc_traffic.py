#!/usr/bin/env python3
"""HBM traffic per launch of the dominant (trunk block) kernel from two rocprofv3 --pmc passes over a short bench.py run
(FETCH_SIZE and WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md "rocprofv3 PMC slots"), corrected as that guide
prescribes (FETCH_SIZE x2 on gfx950, KiB units), stamped with the hash of the kernel sources it was measured on, and
merged into profiles/<round>_pmc_traffic.json (round = $VQAE_ROUND, default r03) under the key <config>_<dtype>_B<batch> that bench.py looks up.

    python tools/pmc_traffic.py --config B --dtype f32 --batch 256          (on the GPU box)
"""
import argparse, csv, glob, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one_pass(counter, bench_args, tag):
    out = os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_{counter}")
    subprocess.call(["rm", "-rf", out])
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["timeout", "-k", "10", "300", "rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out,
           "--", sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-other-configs"] + bench_args
    subprocess.check_call(cmd, cwd="/tmp", env=env, stdout=open(out + ".log", "w"), stderr=subprocess.STDOUT)
    rows = []
    for f in glob.glob(out + "/*/*_counter_collection.csv"):
        rows += [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    return rows


def load_pass(counter, tag):
    rows = []
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_{counter}", "*", "*_counter_collection.csv")):
        rows += [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reuse", action="store_true", help="summarise the CSVs of an earlier run under gpurun_out/ (no GPU)")
    ap.add_argument("--config", default="B"); ap.add_argument("--dtype", default="f32"); ap.add_argument("--batch", type=int, default=256)
    a = ap.parse_args()
    bench_args = ["--config", a.config, "--dtype", a.dtype, "--batch", str(a.batch)]
    cch = 256 if a.config == "C" else 128                             # trunk channels: the dominant kernel's instantiation
    w43 = os.environ.get("VQAE_WINO43", "1") != "0"                  # round 3: the fp32 trunk runs conv_wino43.hip unless switched off
    kern = ("wino43_trunk_kernel" if w43 else "wino_trunk_kernel") if a.dtype == "f32" else "trunk16_kernel"
    pat = f"{kern}<{cch}, "
    sources = {"wino43_trunk_kernel": ["conv_wino43.hip", "common.h"], "wino_trunk_kernel": ["conv_wino.hip", "common.h"],
               "trunk16_kernel": ["trunk16.hip", "common.h"]}[kern]
    tot = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        rows = [r for r in (load_pass(counter, f"{a.config}_{a.dtype}") if a.reuse else one_pass(counter, bench_args, f"{a.config}_{a.dtype}"))
                if pat in r["Kernel_Name"]]
        big = max(int(r["Grid_Size"]) for r in rows)
        rows = [r for r in rows if int(r["Grid_Size"]) == big]          # full-batch launches only (not the calibration batch)
        tot[counter] = sum(float(r["Counter_Value"]) for r in rows) / len(rows) * 1024.0     # KiB -> bytes
        tot["launches"] = len(rows)
    from bench import kernel_source_hash
    rnd = os.environ.get("VQAE_ROUND", "r03")
    path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_traffic.json")
    d = json.load(open(path)) if os.path.exists(path) else {}
    d[f"{a.config}_{a.dtype}_B{a.batch}"] = {
        "kernel": pat.rstrip(", ") + ", ...>", "fetch_size_bytes_raw": tot["FETCH_SIZE"], "write_size_bytes": tot["WRITE_SIZE"],
        "bytes_per_launch": 2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"], "launches_averaged": tot["launches"],
        "correction": "FETCH_SIZE x2 (gfx950 reports half of wide coalesced reads), KiB -> bytes; WRITE_SIZE as is",
        "sources": sources, "source_hash": kernel_source_hash(sources)}
    json.dump(d, open(path, "w"), indent=1)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(d, open(os.path.join(ROOT, "gpurun_out", f"{rnd}_pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(d[f"{a.config}_{a.dtype}_B{a.batch}"]))


if __name__ == "__main__":
    main()
