#!/usr/bin/env python
"""Per-kernel mean / median of the counters collected by tools/collect_profiles.sh
(rocprofv3 counter_collection.csv files under <dir>/pmc*/)."""
import csv
import glob
import json
import os
import re
import statistics
import sys


def short(name):
    name = re.sub(r"^void\s+", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.strip()


def main():
    root = sys.argv[1]
    acc = {}
    for path in glob.glob(os.path.join(root, "pmc*", "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = short(row.get("Kernel_Name", ""))
                c = row.get("Counter_Name")
                try:
                    v = float(row.get("Counter_Value", "nan"))
                except ValueError:
                    continue
                # one row per (dispatch, counter[, dimension]): sum the dimensions of a dispatch
                d = (path, row.get("Dispatch_Id"))
                acc.setdefault(k, {}).setdefault(c, {}).setdefault(d, 0.0)
                acc[k][c][d] += v
    out = {}
    for k, cs in sorted(acc.items()):
        out[k] = {}
        for c, per in sorted(cs.items()):
            vals = list(per.values())
            out[k][c] = {"launches": len(vals), "mean": statistics.fmean(vals), "median": statistics.median(vals), "max": max(vals)}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
