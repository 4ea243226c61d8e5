#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel trace + counter collection) into a small JSON/markdown that is committed
under profiles/. Usage: summarize_pmc.py <rocprof_dir> [<rocprof_dir> ...] > summary.json"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    out = {"kernels": {}, "counters": {}}
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            durs = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"].split("(")[0]
                durs[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
            for k, v in durs.items():
                e = out["kernels"].setdefault(k, {"calls": 0, "total_ms": 0.0})
                e["calls"] += len(v)
                e["total_ms"] += sum(v)
                e["avg_ms"] = e["total_ms"] / e["calls"]
                e["max_ms"] = max(e.get("max_ms", 0.0), max(v))
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"].split("(")[0]
                agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
            for (k, c), v in agg.items():
                out["counters"].setdefault(k, {})[c] = {"dispatches": len(v), "mean": sum(v) / len(v), "last": v[-1]}
    json.dump(out, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
