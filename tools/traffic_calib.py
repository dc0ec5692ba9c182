#!/usr/bin/env python
"""Counter bytes per access of tools/traffic_probe's kernels (see tools/traffic_calib.sh)."""
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r"^void\s+", "", name)
    return re.sub(r"\(.*$", "", name).strip()


def main():
    root = sys.argv[1]
    probe = {}
    for line in open(os.path.join(root, "probe.log")):
        if line.startswith("{"):
            d = json.loads(line)
            probe[d["kernel"]] = d
    tally = {}
    for path in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                k, c = short(row.get("Kernel_Name", "")), row.get("Counter_Name")
                try:
                    v = float(row.get("Counter_Value", "nan"))
                except ValueError:
                    continue
                tally.setdefault(k, {}).setdefault(c, 0.0)
                tally[k][c] += v
    out = {}
    for k, p in probe.items():
        t = tally.get(k, {})
        n = p["accesses"]
        e = {"accesses": n, "useful_bytes_per_access": p["useful_bytes"] / n, "ms": p["ms"],
             "G_accesses_per_s": n / p["ms"] / 1e6, "useful_GB_per_s": p["useful_bytes"] / p["ms"] / 1e6}
        for c, v in sorted(t.items()):
            e[c] = v
            if c in ("FETCH_SIZE", "WRITE_SIZE"):
                e[c + "_bytes_per_access"] = v * 1024.0 / n          # (the derived metrics are in KiB)
            else:
                e[c + "_per_access"] = v / n
        out[k] = e
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
