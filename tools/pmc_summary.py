#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection.csv files per kernel: sum of each counter and launch count.

    python tools/pmc_summary.py out.json dir_or_csv [dir_or_csv ...]
"""
import csv
import glob
import json
import os
import re
import sys


def short(name):
    m = re.match(r"(?:void )?(?:edev::)?([A-Za-z_0-9]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name


def main():
    out, srcs = sys.argv[1], sys.argv[2:]
    agg = {}
    for src in srcs:
        files = [src] if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
        for f in files:
            seen = {}
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                c = r["Counter_Name"]
                d = agg.setdefault(k, {})
                d[c] = d.get(c, 0.0) + float(r["Counter_Value"])
                seen.setdefault((k, c), set()).add(r["Dispatch_Id"])
            for (k, c), ids in seen.items():
                agg[k]["launches"] = max(agg[k].get("launches", 0), len(ids))
    json.dump(agg, open(out, "w"), indent=1, sort_keys=True)
    for k, d in sorted(agg.items()):
        print(k, {c: ("%.4g" % v) for c, v in sorted(d.items())})


if __name__ == "__main__":
    main()
