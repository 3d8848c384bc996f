#!/usr/bin/env python3
"""profiles/traffic.json from the per-kernel tables of tools/pmc_passes.sh (tools/pmc_table.py output).

One entry per workload, keyed as bench.py keys its workloads ("<variant>_<integrator>_<light strategy>").  Per kernel
family (closest = k_intersect_closest, shadow = the shadow-ray kernel of the workload, shade = the largest k_shade kernel):
HBM bytes per launch ((2 * FETCH_SIZE + WRITE_SIZE) KB over the launches of one 16-spp step: rocprofv3 reports both
counters in KB, and on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so reads are doubled), L2 hit rate, share
of VALU issue slots used (a wave64 instruction every two cycles per SIMD), lanes active per VALU instruction, and the split
of the waves' time into waiting on memory, waiting to issue and issuing.  --commit names the tree the passes were taken on; bench.py
prints it beside the figures it reads from here.

    python tools/traffic_json.py profiles/traffic.json --commit <hash> microfacet_path_power=profiles/<cornell table>.json ...
"""
import json
import sys

args = sys.argv[1:]
dst = args.pop(0)
commit = None
if args and args[0] == "--commit":
    commit = args[1]
    args = args[2:]
out = {"unit": "bytes per launch", "note": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 / launches of one 16-spp step at 1080p, separate --pmc passes"}


def row_of(table, name):
    r = table[name]
    row = {"kernel": name, "bytes_per_launch": r["hbm_GB"] * 1e9 / r["launches"], "ms_per_launch": r["ms_per_step"] / r["launches"], "l2_hit": r["l2_hit_rate"],
           "valu_issue_share": r["valu_issue_share"], "valu_lane_use": r["valu_lane_use"]}
    for k in ("wave_time", "clock_GHz", "l1_to_l2_read_latency_cycles"):   # where the waves' time goes (pmc_table.py)
        if r.get(k) is not None:
            row[k] = r[k]
    return row


for arg in args:
    key, src = arg.split("=", 1)
    table = json.load(open(src))
    kernels = {}
    closest = next((k for k in table if k.startswith("k_intersect_closest")), None)
    shadow = next((k for k in table if k.startswith(("k_shadow_stream", "k_shadow_tr", "k_shadow<"))), None)
    shades = [k for k in table if k.startswith(("k_shade<", "k_shade_direct<", "k_shade_medium"))]
    if closest:
        kernels["closest"] = row_of(table, closest)
    if shadow:
        kernels["shadow"] = row_of(table, shadow)
    if shades:
        # the shading kernels of a step together: bytes and time summed, issue share and lane use weighted by time
        t = sum(table[k]["ms_per_step"] for k in shades)
        launches = max(table[k]["launches"] for k in shades)
        kernels["shade"] = {"kernel": " + ".join(shades), "bytes_per_launch": sum(table[k]["hbm_GB"] for k in shades) * 1e9 / launches, "ms_per_launch": t / launches,
                            "l2_hit": sum(table[k]["l2_hit_rate"] * table[k]["ms_per_step"] for k in shades) / t,
                            "valu_issue_share": sum(table[k]["valu_issue_share"] * table[k]["ms_per_step"] for k in shades) / t,
                            "valu_lane_use": sum(table[k]["valu_lane_use"] * table[k]["ms_per_step"] for k in shades) / t}
        if all(table[k].get("wave_time") for k in shades):
            kernels["shade"]["wave_time"] = {f: round(sum(table[k]["wave_time"][f] * table[k]["ms_per_step"] for k in shades) / t, 2) for f in table[shades[0]]["wave_time"]}
    out[key] = {"commit": commit, "source": src, "kernels": kernels}
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
