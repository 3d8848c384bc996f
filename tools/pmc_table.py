"""Per-kernel table from the passes of tools/pmc_passes.sh: time, HBM bytes, L2 hit rate, VALU issue share and lane use, where
the waves' time goes, and the vector cache's average read latency to L2.

VALU issue share = SQ_INSTS_VALU / (CUs * 4 SIMDs * kernel cycles / 2): a SIMD of gfx950 finishes a wave64 VALU instruction
every two cycles when it has waves to take them from (one wave alone issues one every four).  Kernel cycles =
GRBM_GUI_ACTIVE / 8 of the same kernel (the counter is summed over the eight XCDs), i.e. at the clock the chip really ran;
without that pass, trace time x 2.4 GHz.  Lane use = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU).
Wave time: SQ_WAIT_ANY (parked on s_waitcnt: memory, LDS, barriers), SQ_WAIT_INST_ANY (ready, waiting for an issue slot
or an operand), SQ_ACTIVE_INST_ANY (issuing), each over SQ_WAVE_CYCLES.  HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) KB
(gfx950 counts 128-byte reads as 64)."""
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
c = {d: counters(d) for d in ("fetch", "write", "l2", "sq1", "sq2", "sqw", "tcp")}
table = {}
for k in sorted(ms, key=lambda k: -ms[k]):
    if "rocclr" in k or "k_stage" in k:
        continue
    t = ms[k]
    hbm = (2 * c["fetch"][k].get("FETCH_SIZE", 0) + c["write"][k].get("WRITE_SIZE", 0)) * 1024
    hit, miss = c["l2"][k].get("TCC_HIT_sum", 0), c["l2"][k].get("TCC_MISS_sum", 0)
    valu, act, thr = c["sq1"][k].get("SQ_INSTS_VALU", 0), c["sq1"][k].get("SQ_ACTIVE_INST_VALU", 0), c["sq1"][k].get("SQ_THREAD_CYCLES_VALU", 0)
    gui = c["sq2"][k].get("GRBM_GUI_ACTIVE", 0)
    cycles = gui / 8 if gui else t * 1e-3 * GHZ * 1e9
    slots = CUS * 4 * cycles / 2
    w = c["sqw"][k]
    wc = w.get("SQ_WAVE_CYCLES", 0)
    lat_n, lat_s = c["tcp"][k].get("TCP_TCC_READ_REQ_sum", 0), c["tcp"][k].get("TCP_TCC_READ_REQ_LATENCY_sum", 0)
    table[k] = {"ms_per_step": round(t, 3), "launches": launches[k], "hbm_GB": round(hbm / 1e9, 3), "hbm_TBps": round(hbm / (t * 1e-3) / 1e12, 2) if t else None,
                "l2_hit_rate": round(hit / (hit + miss), 3) if hit + miss else None, "valu_issue_share": round(valu / slots, 2) if slots else None,
                "valu_lane_use": round(thr / (64 * act), 2) if act else None,
                "clock_GHz": round(gui / 8 / (t * 1e-3) / 1e9, 2) if gui and t else None,
                "wave_time": {"waiting_on_memory_or_barrier": round(w.get("SQ_WAIT_ANY", 0) / wc, 2), "waiting_to_issue": round(w.get("SQ_WAIT_INST_ANY", 0) / wc, 2),
                              "issuing": round(w.get("SQ_ACTIVE_INST_ANY", 0) / wc, 2)} if wc else None,
                "l1_to_l2_read_latency_cycles": round(lat_s / lat_n) if lat_n else None}
print(json.dumps(table, indent=1))
