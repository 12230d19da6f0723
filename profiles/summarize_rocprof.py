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
if "--out" in sys.argv:                 # write the summaries somewhere else (on the GPU box: under gpurun_out/)
    i = sys.argv.index("--out")
    HERE = sys.argv[i + 1]
    del sys.argv[i:i + 2]


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
            for r in rows[:90]:
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


def traffic(fetch_tag, write_tag, out_tag):
    """HBM bytes per launch from two separate PMC passes (MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE
    count KiB; on gfx950 FETCH_SIZE tallies 128-B requests as 64 B -> x2)."""
    f = json.load(open(os.path.join(HERE, f"pmc_{fetch_tag}.json")))
    w = json.load(open(os.path.join(HERE, f"pmc_{write_tag}.json")))
    out = {}
    for k in sorted(set(f) | set(w)):
        n = int((f.get(k) or w.get(k))["dispatches"])
        out[k] = {"launches": n,
                  "fetch_bytes_per_launch": int(f.get(k, {}).get("FETCH_SIZE", 0) * 1024 * 2 / max(n, 1)),
                  "write_bytes_per_launch": int(w.get(k, {}).get("WRITE_SIZE", 0) * 1024 / max(n, 1))}
    json.dump({"kernels": out}, open(os.path.join(HERE, f"traffic_{out_tag}.json"), "w"), indent=1, sort_keys=True)
    print("wrote traffic for", len(out), "kernels")


if __name__ == "__main__":
    if sys.argv[1] == "traffic":
        traffic(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        main(sys.argv[1], sys.argv[2])
