#!/usr/bin/env python3
"""Rewrites the measured columns of the measurement tables in the design document at the repository root from a kept set
(tools/keep_set.sh <tag> <commit>).
usage: python tools/design_tables.py <tag>      (reads profiles/<tag>_*, edits the document in place, prints what changed)"""
import json, os, re, sys

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
tag = sys.argv[1]
path = os.path.join(root, "DESIGN.md")
text = open(path).read()


def line_of(name):
    return json.loads(open(os.path.join(root, "profiles", "%s_bench_%s.json" % (tag, name))).read().strip().splitlines()[-1])


def thousands(v):
    v = int(round(v))
    return "{:,}".format(v) if v >= 10000 else str(v)


# workload table: | label | ms/step | Mrays/s | Msamples/s | whole job | earlier value | file |
rows = 0
for name in ("default", "bigmesh_path", "manylights_path", "manylights_path_spatial", "manylights_direct-one", "microfacet_direct-all", "fog_volpath", "bigmesh_fog_volpath"):
    m = re.search(r"^\|[^\n]*`r03_\w+_bench_%s\.json`[^\n]*$" % re.escape(name), text, re.M)
    if not m:
        print("no row for", name)
        continue
    d = line_of(name)
    cells = m.group(0).split("|")
    ms = d["ms_per_step"]
    cells[2] = " %s " % (("%.2f" % ms) if ms < 50 else ("%.1f" % ms))
    cells[3] = " %s " % thousands(d["value"])
    cells[4] = " %d " % round(d["msamples_per_s"])
    job = re.sub(r"\*?\*?[\d.]+ s", lambda mm: mm.group(0).replace(re.search(r"[\d.]+", mm.group(0)).group(0), "%.2f" % d["job_seconds"]), cells[5], count=1)
    cells[5] = job
    cells[7] = re.sub(r"r03_\w+?_bench_", "%s_bench_" % tag, cells[7])
    new = "|".join(cells)
    if new != m.group(0):
        rows += 1
    text = text[:m.start()] + new + text[m.end():]
print("workload rows rewritten:", rows)

# per-kernel tables
short = {"k_shade<0": "k_shade<matte>", "k_shade<3": "k_shade<rough glass>", "k_shade<1": "k_shade<metal>"}
for header, workload in (("| kernel (BigMesh) |", "bigmesh"), ("| kernel (Cornell) |", "cornell"), ("| kernel (BigMesh in fog, VolPath) |", "bigmesh_fog")):
    d = json.load(open(os.path.join(root, "profiles", "%s_pmc_%s_per_kernel.json" % (tag, workload))))
    i = text.index(header)
    j = text.index("\n", text.index("|---|", i)) + 1
    k = j
    out = []
    while text[k] == "|":
        e = text.index("\n", k) + 1
        row = text[k:e]
        label = re.match(r"\| `([^`]+)`", row).group(1)
        want = next((p for p, n in short.items() if n == label), label)
        keys = [x for x in d if x == want or (x.startswith(want) and x[len(want):len(want) + 1] in "<,")]
        if not keys:
            out.append(row)
        else:
            v = d[keys[0]]
            w = v["wave_time"]
            cells = row.split("|")
            ms = v["ms_per_step"]
            cells[2] = " %s " % (("%.1f" % ms) if workload == "bigmesh_fog" else ("%.2f" % ms))
            cells[3] = " %.1f " % v["hbm_GB"]
            cells[4] = " %.1f " % v["hbm_TBps"]
            cells[5] = " %.2f " % v["l2_hit_rate"]
            cells[6] = " %.2f " % v["valu_issue_share"]
            cells[7] = " %.2f " % v["valu_lane_use"]
            cells[8] = " %.2f / %.2f / %.2f " % (w["waiting_on_memory_or_barrier"], w["waiting_to_issue"], w["issuing"])
            cells[9] = " %d " % v["l1_to_l2_read_latency_cycles"]
            out.append("|".join(cells))
        k = e
    text = text[:j] + "".join(out) + text[k:]
open(path, "w").write(text)
d = line_of("default")
print("cornell cpu %.1f Mrays/s (%.0f s), bigmesh cpu %.2f Mrays/s (%.0f s), job %.2f s" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["seconds"],
      d["bigmesh"]["cpu_baseline"]["value"], d["bigmesh"]["cpu_baseline"]["seconds"], d["bigmesh"]["job_seconds"]))
