"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a small text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def rows(path):
    with open(path, newline="") as f:
        return list(csv.DictReader(f))


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for p in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in rows(p):
        print("%-90s calls=%s total_ns=%s avg_ns=%s pct=%s" % (r.get("Name", "")[:90], r.get("Calls"),
              r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage")))
print()
print("== PMC (per kernel name: mean counter value per dispatch) ==")
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    acc = defaultdict(lambda: defaultdict(list))
    for p in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in rows(p):
            name = r.get("Kernel_Name", "")[:70]
            acc[name][r.get("Counter_Name")].append(float(r.get("Counter_Value", 0)))
    print("--", os.path.basename(d))
    for name, cs in acc.items():
        for c, v in cs.items():
            print("   %-70s %-28s n=%d mean=%.6g" % (name, c, len(v), sum(v) / len(v)))
