#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 rocpd SQLite database (`rocprofv3 --kernel-trace -d <dir>` writes
<dir>/<host>/<pid>_results.db when no --output-format is given) -> CSV like `--stats` prints.

    python tools/rocpd_stats.py gpurun_out/prof2 [-o profiles/x.csv] [--last N]   (--last: only the last N dispatches)
"""
import argparse
import collections
import csv
import glob
import re
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([\w:]+(<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:90]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("-o")
    ap.add_argument("--last", type=int, default=0)
    a = ap.parse_args()
    db = sorted(glob.glob(a.dir + "/*/*.db") + glob.glob(a.dir + "/*.db"))[0]
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "info_kernel_symbol" in t][0]
    rows = list(c.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, s.arch_vgpr_count, s.accum_vgpr_count, "
                          f"d.group_segment_size from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
    if a.last:
        rows = rows[-a.last:]
    agg = collections.OrderedDict()
    for name, st, en, grid, vg, ag, lds in rows:
        k = short(name)
        e = agg.setdefault(k, dict(calls=0, total=0, mn=1 << 62, mx=0, vgpr=vg, agpr=ag, lds=lds))
        d = en - st
        e["calls"] += 1; e["total"] += d; e["mn"] = min(e["mn"], d); e["mx"] = max(e["mx"], d)
    tot = sum(e["total"] for e in agg.values()) or 1
    out = csv.writer(open(a.o, "w", newline="") if a.o else sys.stdout)
    out.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "VGPR", "AGPR", "LDS"])
    for k, e in sorted(agg.items(), key=lambda kv: -kv[1]["total"]):
        out.writerow([k, e["calls"], e["total"], round(e["total"] / e["calls"], 1), round(100.0 * e["total"] / tot, 2),
                      e["mn"], e["mx"], e["vgpr"], e["agpr"], e["lds"]])


if __name__ == "__main__":
    main()
