"""Per-kernel table from the passes of tools/pmc_passes.sh: time, HBM bytes, L2 hit rate, VALU issue share and lane use.

VALU issue share = SQ_INSTS_VALU / (CUs * 4 SIMDs * kernel cycles / 4): a wave64 VALU instruction occupies its SIMD for four
cycles.  Lane use = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU).  HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) KB (gfx950
counts 128-byte reads as 64)."""
import csv, glob, json, os, re, sys, collections

out = sys.argv[1]
CUS, GHZ = 256, 2.4


def short(n):
    return re.sub(r"\(.*", "", n).replace("void ", "").replace("edev::", "")


def counters(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(os.path.join(out, d, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
    return agg


ms, launches = collections.Counter(), collections.Counter()
for f in glob.glob(os.path.join(out, "trace", "*kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        ms[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        launches[k] += 1
c = {d: counters(d) for d in ("fetch", "write", "l2", "sq1", "sq2")}
table = {}
for k in sorted(ms, key=lambda k: -ms[k]):
    if "rocclr" in k or "k_stage" in k:
        continue
    t = ms[k]
    hbm = (2 * c["fetch"][k].get("FETCH_SIZE", 0) + c["write"][k].get("WRITE_SIZE", 0)) * 1024
    hit, miss = c["l2"][k].get("TCC_HIT_sum", 0), c["l2"][k].get("TCC_MISS_sum", 0)
    valu, act, thr = c["sq1"][k].get("SQ_INSTS_VALU", 0), c["sq1"][k].get("SQ_ACTIVE_INST_VALU", 0), c["sq1"][k].get("SQ_THREAD_CYCLES_VALU", 0)
    slots = CUS * 4 * (t * 1e-3 * GHZ * 1e9) / 4
    table[k] = {"ms_per_step": round(t, 3), "launches": launches[k], "hbm_GB": round(hbm / 1e9, 3), "hbm_TBps": round(hbm / (t * 1e-3) / 1e12, 2) if t else None,
                "l2_hit_rate": round(hit / (hit + miss), 3) if hit + miss else None, "valu_issue_share": round(valu / slots, 2) if slots else None,
                "valu_lane_use": round(thr / (64 * act), 2) if act else None}
print(json.dumps(table, indent=1))
