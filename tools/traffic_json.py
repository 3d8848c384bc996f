#!/usr/bin/env python3
"""profiles/traffic.json from the per-kernel tables of tools/pmc_passes.sh (tools/pmc_table.py output).

One entry per workload, keyed as bench.py keys its workloads ("<variant>_<integrator>_<light strategy>"): the HBM bytes
per launch of k_intersect_closest ((2 * FETCH_SIZE + WRITE_SIZE) KB over the launches of one 16-spp step: rocprofv3
reports both counters in KB, and on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so reads are doubled), and
that kernel's L2 hit rate, share of VALU issue slots and lanes active per VALU instruction.

    python tools/traffic_json.py profiles/traffic.json microfacet_path_power=profiles/<cornell table>.json bigmesh_path_power=profiles/<bigmesh table>.json
"""
import json
import sys

dst = sys.argv[1]
out = {"unit": "bytes per launch", "note": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 / launches of one 16-spp step at 1080p, separate --pmc passes"}
for arg in sys.argv[2:]:
    key, src = arg.split("=", 1)
    table = json.load(open(src))
    name = next(k for k in table if k.startswith("k_intersect_closest"))
    row = table[name]
    out[key] = {"k_intersect_closest_bytes_per_launch": row["hbm_GB"] * 1e9 / row["launches"], "l2_hit": row["l2_hit_rate"],
                "valu_issue_share": row["valu_issue_share"], "valu_lane_use": row["valu_lane_use"], "kernel": name, "source": src}
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
