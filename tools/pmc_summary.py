"""Per-kernel means of the PMC passes written by tools/gpu_pmc.sh.
    python tools/pmc_summary.py gpurun_out/<tag> [substring filter]   -> JSON on stdout
rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB; on gfx950 FETCH_SIZE counts 64 B per 128-B request
for wide coalesced reads (MI355X_MICROARCH.md, HBM section): `fetch_bytes_corrected` doubles it.
"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(root + "/*/**/*counter_collection.csv", recursive=True):
    disp = defaultdict(dict)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if flt and flt not in name:
            continue
        key = (name, r["Dispatch_Id"])
        disp[key][r["Counter_Name"]] = disp[key].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for (name, _), cs in disp.items():
        for c, v in cs.items():
            acc[name][c].append(v)
out = {}
for name, cs in acc.items():
    m = re.search(r"(knn_\w+|ransac_\w+|filter_\w+|concat_\w+)(<[^(]*>)?", name)
    short = m.group(0) if m else name[:60]
    d = {c: sum(v[len(v) // 2:]) / max(1, len(v[len(v) // 2:])) for c, v in cs.items()}   # later half: warmed up
    if "FETCH_SIZE" in d:
        d["fetch_bytes_corrected"] = d["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in d:
        d["write_bytes"] = d["WRITE_SIZE"] * 1024
    d["launches_seen"] = max(len(v) for v in cs.values())
    out[short] = d
json.dump(out, sys.stdout, indent=1)
