#!/usr/bin/env python
"""Condenses rocprofv3 CSV output (kernel stats / counter collection) into small committed
summaries.  Usage: python profiles/summarize_rocprof.py <rocprof_out_dir> <tag>
Writes profiles/rocprof_<tag>_kernel_stats.md and, if counters were collected,
profiles/pmc_<tag>.json (per-kernel sums of each counter and dispatch counts)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

HERE = os.path.dirname(os.path.abspath(__file__))


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void swn::", "").replace("swn::", "")
    name = re.sub(r"\(.*", "", name)
    return name[:90]


def main(src, tag):
    stats = glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        rows.sort(key=lambda r: -float(r.get("TotalDurationNs", r.get("Total", 0)) or 0))
        tot = sum(float(r["TotalDurationNs"]) for r in rows) or 1.0
        with open(os.path.join(HERE, f"rocprof_{tag}_kernel_stats.md"), "w") as f:
            f.write(f"# rocprofv3 --kernel-trace --stats ({tag})\n\n")
            f.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
            for r in rows[:40]:
                f.write("| %s | %s | %.3f | %.1f | %.2f |\n" % (
                    short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                    float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
        print("wrote kernel stats for", len(rows), "kernels")
    cnt = glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
    if cnt:
        agg = defaultdict(lambda: defaultdict(float))
        disp = defaultdict(set)
        for path in cnt:
            for r in csv.DictReader(open(path)):
                k = short(r["Kernel_Name"])
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                disp[k].add(r["Dispatch_Id"])
        out = {k: dict(v, dispatches=len(disp[k])) for k, v in agg.items()}
        json.dump(out, open(os.path.join(HERE, f"pmc_{tag}.json"), "w"), indent=1, sort_keys=True)
        print("wrote counters for", len(out), "kernels")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
