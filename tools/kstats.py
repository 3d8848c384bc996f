"""Per-kernel totals (ms) from a rocprofv3 kernel-trace CSV, divided by a step count."""
import csv, sys, collections, re
path, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot, cnt = collections.Counter(), collections.Counter()
for r in csv.DictReader(open(path)):
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    tot[name] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    cnt[name] += 1
for name, ms in tot.most_common():
    print(f"{ms / steps:9.3f} ms/step  {cnt[name]:5d} launches  {name}")
print(f"{sum(tot.values()) / steps:9.3f} ms/step  total")
