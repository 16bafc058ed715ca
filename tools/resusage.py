#!/usr/bin/env python
"""Per-kernel resource table from the compiler's -Rpass-analysis=kernel-resource-usage report (csrc/resource_usage.txt).
usage: python tools/resusage.py [substring]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
txt = open(os.path.join(ROOT, "smc.jl_amd", "csrc", "resource_usage.txt")).read()
flt = sys.argv[1] if len(sys.argv) > 1 else ""
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
names = [b.split("\n")[0].strip() for b in blocks]
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.strip().split("\n")
keys = [("vgpr", r"VGPRs"), ("agpr", r"AGPRs"), ("sgpr", r"SGPRs"), ("scratch", r"ScratchSize \[bytes/lane\]"), ("occ", r"Occupancy \[waves/SIMD\]"),
        ("lds", r"LDS Size \[bytes/block\]")]
for b, dn in zip(blocks, dem):
    if flt not in dn:
        continue
    vals = []
    for k, pat in keys:
        m = re.search(pat + r": (\d+)", b)
        vals.append("%s=%s" % (k, m.group(1) if m else "?"))
    print("%-64s %s" % (dn.split("(")[0][:64], " ".join(vals)))
