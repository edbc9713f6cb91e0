"""Turn one measurement session (gpurun_out/<tag>, written by tools/gpu_round3.sh) into the committed artefacts under
profiles/: bench lines, rocprofv3 kernel stats, per-kernel durations recomputed from the trace, PMC summaries and the
source-stamped traffic file bench.py reads.       python tools/collect_profiles.py gpurun_out/<tag> [r02]"""
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src = sys.argv[1]
pre = sys.argv[2] if len(sys.argv) > 2 else "r03"
P = os.path.join(ROOT, "profiles")


def last_json_line(path):
    lines = [ln for ln in open(path).read().strip().splitlines() if ln.startswith("{")]
    return json.loads(lines[-1])


# ---- bench lines
for f in sorted(glob.glob(os.path.join(src, "bench_*.json"))):
    d = last_json_line(f)
    json.dump(d, open(os.path.join(P, "%s_%s" % (pre, os.path.basename(f))), "w"), indent=1)
for extra in ("fallback_perf.json",):
    if os.path.exists(os.path.join(src, extra)):
        json.dump(last_json_line(os.path.join(src, extra)), open(os.path.join(P, "%s_%s" % (pre, extra)), "w"), indent=1)
for txt_name, head in (("mgpu_stream.txt", "tools/mgpu_stream_bench.py: streamed multi-GPU entry points with the one device of the box"),
                       ("flann.txt", "tools/prof_flann.py: kd-forest search, one wave per query"),
                       ("sweep_ratio_8192.txt", "tools/sweep_ratio.py 8192 8192: matcher call by route and option"),
                       ("sweep_u8_forms_32k.txt", "tools/sweep_u8.py 32768 32768: the six forms of the u8 coarse kernel (PM_OPT_KNN_RING)"),
                       ("sweep_u8_forms_16k.txt", "tools/sweep_u8.py 16384 16384: tile kernel and register-operand forms 5, 6"),
                       ("knn_stamps_32k.txt", "tools/prof_knn_stamps.py 32768 32768 on the stamping build: ring form (12=2) and split-per-wave form (12=6)"),
                       ("ransac_stamps.txt", "tools/prof_ransac_stamps.py on the stamping build: phase timeline of the one-launch RANSAC kernel, both forms")):
    if os.path.exists(os.path.join(src, txt_name)):
        body = [ln for ln in open(os.path.join(src, txt_name)) if "amdgpu.ids" not in ln]
        open(os.path.join(P, "%s_%s" % (pre, txt_name)), "w").write("# " + head + "\n" + "".join(body))
if os.path.exists(os.path.join(src, "h2d_ceiling.txt")):
    txt = [ln for ln in open(os.path.join(src, "h2d_ceiling.txt")) if "amdgpu.ids" not in ln]
    open(os.path.join(P, "%s_h2d_ceiling.txt" % pre), "w").write(
        "# tools/h2d_ceiling.py on the GPU box: pinned host -> device copy rate by chunk size and stream count\n" + "".join(txt))

# ---- kernel traces
for cfg in ("c3", "c4", "c3_surf", "l32k"):
    # (a re-run leaves the earlier run's files next to the new ones: newest first)
    st = sorted(glob.glob(os.path.join(src, "trace_" + cfg, "*", "*kernel_stats.csv")), key=os.path.getmtime, reverse=True)
    tr = sorted(glob.glob(os.path.join(src, "trace_" + cfg, "*", "*kernel_trace.csv")), key=os.path.getmtime, reverse=True)
    if st:
        shutil.copy(st[0], os.path.join(P, "%s_bench_%s_kernel_stats.csv" % (pre, cfg)))
    if tr and cfg in ("c3", "c3_surf", "l32k"):
        per = defaultdict(list)
        for r in csv.DictReader(open(tr[0])):
            per[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        out = {}
        for name, v in per.items():
            m = re.search(r"(knn_\w+|ransac_\w+|filter_\w+|lmeds_\w+)(<[^(]*>)?", name)
            if not m:
                continue
            short = m.group(0)
            counted = v
            if "knn_mfma_rows288" in short or short.startswith("knn_l2_mfma"):   # launches that exit in their first instructions
                counted = [x for x in v if x > 8.0] or v
            out[short] = {"calls": len(v), "calls_counted": len(counted), "avg_us": round(sum(counted) / max(1, len(counted)), 3),
                          "min_us": round(min(counted), 3) if counted else None, "max_us": round(max(counted), 3) if counted else None}
        cmdline = ("tools/prof_knn.py 32768 32768 128 10 sift 8" if cfg == "l32k" else
                   "bench.py%s --steps 20 --warmup 5 --no-cpu-baseline --no-verify --sustain-seconds 0 --headline-only --no-large"
                   % (" --kind surf" if cfg == "c3_surf" else ""))
        json.dump({"note": "per-kernel durations from the rocprofv3 --kernel-trace of `%s` (the *_kernel_stats.csv next to "
                           "this file is rocprofv3's own --stats summary of the same run). Launches of a coarse kernel that "
                           "exit in their first instructions (the automatic route's unused fallback, < 8 us) are excluded from "
                           "`calls_counted`." % cmdline, "kernels": out},
                  open(os.path.join(P, "%s_bench_%s_kernel_durations.json" % (pre, cfg)), "w"), indent=1)

# ---- PMC summaries
summ = {}
for n, name in (("knn", "pmc_c3"), ("knn32k", "pmc_l32k"), ("ransac", "pmc_ransac_c3"), ("ham", "pmc_c4")):
    d = os.path.join(src, "pmc_" + n)
    if not os.path.isdir(d):
        continue
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), d], capture_output=True, text=True)
    summ[n] = json.loads(r.stdout)
    json.dump(summ[n], open(os.path.join(P, "%s_%s.json" % (pre, name)), "w"), indent=1)

# ---- traffic, stamped with the sources it was measured on
import bench  # noqa: E402


def pick(s, needle):
    for k, v in s.items():
        if needle in k and "fetch_bytes_corrected" in v and "write_bytes" in v:
            return v
    return None


kern = {}
a = pick(summ.get("knn", {}), "knn_mfma_rows288")
if a:
    kern["c3:knn_l2_mfma_u8"] = {"workload": "C3 8192x8192x128 sift, u8 route (i8 MFMA on x - 128)", "fetch_bytes": a["fetch_bytes_corrected"],
                                 "write_bytes": a["write_bytes"], "traffic_bytes": a["fetch_bytes_corrected"] + a["write_bytes"]}
a = pick(summ.get("knn32k", {}), "knn_mfma_rows288")
if a:
    kern["l32k:knn_l2_mfma_u8"] = {"workload": "32768x32768x128 sift, u8 route (i8 MFMA on x - 128)", "fetch_bytes": a["fetch_bytes_corrected"],
                                   "write_bytes": a["write_bytes"], "traffic_bytes": a["fetch_bytes_corrected"] + a["write_bytes"]}
b = pick(summ.get("ham", {}), "knn_mfma_rows288")
if b:
    kern["c4:knn_hamming_mfma_i8"] = {"workload": "C4 32768x32768 ORB-256, i8 route", "fetch_bytes": b["fetch_bytes_corrected"],
                                      "write_bytes": b["write_bytes"], "traffic_bytes": b["fetch_bytes_corrected"] + b["write_bytes"]}
json.dump({"note": "HBM-side bytes per launch from the rocprofv3 PMC passes of tools/gpu_round3.sh (FETCH_SIZE and WRITE_SIZE each "
                   "in its own run, --kernel-trace only), means over the warmed-up launches; FETCH_SIZE doubled per the gfx950 "
                   "correction of MI355X_MICROARCH.md (64 B tallied per 128-B request), WRITE_SIZE as reported. bench.py reports "
                   "these as roofline.traffic only while kernel_source_sha16 matches the sources in the tree.",
           "kernel_source_sha16": bench._kernel_source_sha(), "kernels": kern},
          open(os.path.join(P, "%s_traffic.json" % pre), "w"), indent=1)
print("profiles updated from", src, "sha", bench._kernel_source_sha(), "kernels", list(kern))
