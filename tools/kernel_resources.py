"""Compile one csrc translation unit for gfx950 and print one line per kernel: VGPRs, AGPRs, SGPRs, scratch, LDS, occupancy.

    python tools/kernel_resources.py knn_coarse.hip [filter-substring]

(hipcc -Rpass-analysis=kernel-resource-usage; nothing is linked or run.)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from points_matching_amd import build as B  # noqa: E402


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    path = src if os.path.exists(src) else os.path.join(B.CSRC, src)
    cmd = [B.HIPCC] + B.FLAGS + B.EXTRA.get(os.path.basename(src), []) + ["-x", "hip", "-c", path, "-o", "/dev/null",
                                                                            "-Rpass-analysis=kernel-resource-usage"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        print(r.stderr)
        sys.exit(1)
    cur = None
    rows = {}
    for ln in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            cur = m.group(1)
            rows[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z][A-Za-z \[\]/]*):\s+(\S+)", ln)
        if m and cur:
            rows[cur][m.group(1).strip()] = m.group(2)
    for name, d in rows.items():
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"(pm_\w+::)?\(anonymous namespace\)::", "", dem)
        dem = re.sub(r"\(.*$", "", dem)
        if flt and flt not in dem:
            continue
        print("%-90s vgpr %3s agpr %3s sgpr %3s scratch %4s lds %6s occ %s" % (
            dem[:90], d.get("VGPRs", "?"), d.get("AGPRs", "?"), d.get("TotalSGPRs", "?"), d.get("ScratchSize [bytes/lane]", "?"),
            d.get("LDS Size [bytes/block]", "?"), d.get("Occupancy [waves/SIMD]", "?")))


if __name__ == "__main__":
    main()
