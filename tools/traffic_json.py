#!/usr/bin/env python3
"""profiles/traffic.json from a PMC summary (tools/pmc_summary.py output holding FETCH_SIZE and WRITE_SIZE).

HBM bytes per launch of k_intersect_closest = (2 * FETCH_SIZE + WRITE_SIZE) KB / launches: rocprofv3
reports both in KB, and on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so reads are doubled.

    python tools/traffic_json.py profiles/<summary>.json profiles/traffic.json
"""
import json
import sys

src, dst = sys.argv[1], sys.argv[2]
summary = json.load(open(src))
out = {"source": src, "unit": "bytes per launch", "note": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 / launches, separate --pmc passes"}
for name, c in summary.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c and c.get("launches"):
        key = name.split("<")[0]
        out[key + "_bytes_per_launch"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 / c["launches"]
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
