#!/usr/bin/env python
"""Per-kernel resource table from the compiler's -Rpass-analysis=kernel-resource-usage report (csrc/resource_usage.txt).
usage: python tools/resusage.py [substring ...]   (a kernel is listed if its demangled name contains any of the substrings)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
txt = open(os.path.join(ROOT, "smc.jl_amd", "csrc", "resource_usage.txt")).read()
flt = sys.argv[1:] or [""]
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
names = [b.split("\n")[0].split()[0].strip() for b in blocks]
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.strip().split("\n")
keys = [("vgpr", r"VGPRs"), ("agpr", r"AGPRs"), ("sgpr", r"SGPRs"), ("scratch", r"ScratchSize \[bytes/lane\]"), ("occ", r"Occupancy \[waves/SIMD\]"),
        ("sspill", r"SGPRs Spill"), ("vspill", r"VGPRs Spill"), ("lds", r"LDS Size \[bytes/block\]")]
seen = set()
for b, dn in zip(blocks, dem):
    short = re.sub(r"^void ", "", dn).split("(")[0].replace("smcmi::", "")
    if short in seen or not any(f in short for f in flt):
        continue
    seen.add(short)
    vals = []
    for k, pat in keys:
        m = re.search(pat + r": (\d+)", b)
        vals.append("%s=%s" % (k, m.group(1) if m else "?"))
    print("%-48s %s" % (short[:48], " ".join(vals)))
