#!/usr/bin/env python
"""Extract one kernel's gfx950 ISA from `hipcc -S --cuda-device-only` output and count a few instruction classes.
usage: python tools/isa_extract.py smcmi.s <mangled-name-prefix> [out.s]"""
import sys

lines = open(sys.argv[1]).read().split("\n")
pref = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith(pref) and l.rstrip().split(";")[0].rstrip().endswith(":"))
out = []
for l in lines[start:]:
    out.append(l)
    if l.startswith(".Lfunc_end"):
        break
txt = "\n".join(out)
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write(txt)
keys = ["ds_bpermute", "_dpp", "v_permlane", "s_waitcnt", "s_barrier", "v_div_", "v_rcp_f64", "v_sqrt_f64", "v_rsq_f64", "v_fma_f64", "v_mul_f64", "v_add_f64",
        "global_load", "global_store", "scratch_", "ds_read", "ds_write", "v_readlane", "v_exp", "v_log", "s_load"]
print(len(out), "lines;", ", ".join("%s %d" % (k, txt.count(k)) for k in keys))
