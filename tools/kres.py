#!/usr/bin/env python3
"""Compile one .hip source of the library for gfx950 and print a compact table of each kernel's
registers, scratch, LDS and occupancy (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kres.py dt_kernels.hip [filter]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from descriptools_amd.build import CSRC, FLAGS  # noqa: E402


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    cmd = ["/opt/rocm/bin/hipcc"] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", "/tmp/kres.o",
                                            "-Rpass-analysis=kernel-resource-usage"]
    txt = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = None
    rows = []
    for line in txt.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["/usr/bin/c++filt", m.group(1)], capture_output=True,
                                  text=True).stdout.strip()
            cur = {"name": re.sub(r"\(.*", "", name)}
            rows.append(cur)
            continue
        for key, pat in (("sgpr", r"TotalSGPRs: (\d+)"), ("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"),
                         ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                         ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    print("%-60s %5s %5s %7s %4s %7s" % ("kernel", "vgpr", "sgpr", "scratch", "occ", "lds"))
    for r in rows:
        if flt in r["name"]:
            print("%-60s %5d %5d %7d %4d %7d" % (r["name"][:60], r.get("vgpr", -1), r.get("sgpr", -1),
                                                 r.get("scratch", -1), r.get("occ", -1), r.get("lds", -1)))


if __name__ == "__main__":
    main()
