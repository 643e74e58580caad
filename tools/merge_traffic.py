"""Fold the traffic.json of one or more tools/gpu_profile.sh runs into profiles/traffic_latest.json
(entries keyed "<workload>/<precision>"; bench.py quotes an entry only for the kernel sources it was
measured on).  usage: python tools/merge_traffic.py gpurun_out/prof_<tag>/traffic.json ..."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(ROOT, "profiles", "traffic_latest.json")
try:
    cur = json.load(open(dst))
except Exception:  # noqa: BLE001
    cur = {}
entries = cur.get("entries", {})
for p in sys.argv[1:]:
    if not os.path.exists(p):
        continue
    e = json.load(open(p))
    if "per_launch_bytes" in e and "workload" in e:
        entries["%s/%s" % (e["workload"], e["precision"])] = e
json.dump({"entries": entries}, open(dst, "w"), indent=1)
print("profiles/traffic_latest.json:", ", ".join(sorted(entries)))
