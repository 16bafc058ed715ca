#!/usr/bin/env python
"""Static instruction mix of the stage kernels (gfx950 ISA from `hipcc -S`), written to profiles/r01_isa_counts.json.
Used by bench.py to express the mutation kernel's speed as a fraction of the VALU issue rate (its real bound - see DESIGN §6).
usage: python profiles/isa_count.py"""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "smc.jl_amd", "csrc", "smcmi.hip")
OUT = os.path.join(ROOT, "profiles", "r01_isa_counts.json")
asm = "/tmp/smcmi_isa.s"
def compile_lines(extra):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S", "--cuda-device-only",
                           "-o", asm, SRC] + extra, stderr=subprocess.DEVNULL)
    return open(asm).read().split("\n")


lines = compile_lines([])
want = {"k_mutate_reg<10,true>": "k_mutate_regILi10ELb1E", "k_mutate_reg<9,true>": "k_mutate_regILi9ELb1E",
        "k_mutate_reg<10,false>": "k_mutate_regILi10ELb0E", "k_pass<16,false>": "k_passILi16ELb0E", "k_pass<1,true>": "k_passILi1ELb1E",
        "k_moments_reg<10>": "k_moments_regILi10E"}
res = {}


def count(lines, key):
    start = next((i for i, l in enumerate(lines) if re.match(r"^_ZN5smcmi\d+" + re.escape(key) + r".*:\s*(;.*)?$", l)), None)
    if start is None:
        return None
    end = next(i for i in range(start, len(lines)) if ".Lfunc_end" in lines[i] and lines[i].strip().endswith(":"))
    cnt = collections.Counter()
    for l in lines[start + 1:end]:
        t = l.strip()
        if not t or t.startswith((";", ".")) or t.endswith(":"):
            continue
        cnt[t.split()[0]] += 1
    g = collections.Counter()
    for op, c in cnt.items():
        if op.startswith("v_") and "f64" in op:
            g["valu_f64"] += c
        elif op.startswith(("v_mad_u64", "v_mul_hi", "v_mul_lo")):
            g["valu_int_mul"] += c
        elif op.startswith("v_"):
            g["valu_other"] += c
        elif op.startswith("s_"):
            g["salu"] += c
        elif op.startswith("ds_"):
            g["lds"] += c
        elif op.startswith(("global", "buffer", "flat", "scratch")):
            g["vmem"] += c
    g["valu_total"] = g["valu_f64"] + g["valu_int_mul"] + g["valu_other"]
    g["all"] = sum(cnt.values())
    return dict(g)


for nice, key in want.items():
    r = count(lines, key)
    if r:
        res[nice] = r
# the mutation kernel as it runs with the random numbers drawn ahead (k_prepare_mutation's idle CUs): in-kernel RNG dead-coded
lines2 = compile_lines(["-DSMCMI_COUNT_RNG_AHEAD"])
for nice, key in want.items():
    if "mutate" in nice:
        r = count(lines2, key)
        if r:
            res[nice + " rng_ahead"] = r
json.dump(res, open(OUT, "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
sys.exit(0)
for nice, key in want.items():
    start = next((i for i, l in enumerate(lines) if re.match(r"^_ZN5smcmi\d+" + re.escape(key) + r".*:\s*(;.*)?$", l)), None)
    if start is None:
        continue
    end = next(i for i in range(start, len(lines)) if ".Lfunc_end" in lines[i] and lines[i].strip().endswith(":"))
    cnt = collections.Counter()
    for l in lines[start + 1:end]:
        t = l.strip()
        if not t or t.startswith((";", ".")) or t.endswith(":"):
            continue
        cnt[t.split()[0]] += 1
    g = collections.Counter()
    for op, c in cnt.items():
        if op.startswith("v_") and "f64" in op:
            g["valu_f64"] += c
        elif op.startswith(("v_mad_u64", "v_mul_hi", "v_mul_lo")):
            g["valu_int_mul"] += c
        elif op.startswith("v_"):
            g["valu_other"] += c
        elif op.startswith("s_"):
            g["salu"] += c
        elif op.startswith("ds_"):
            g["lds"] += c
        elif op.startswith(("global", "buffer", "flat", "scratch")):
            g["vmem"] += c
    g["valu_total"] = g["valu_f64"] + g["valu_int_mul"] + g["valu_other"]
    g["all"] = sum(cnt.values())
    res[nice] = dict(g)
json.dump(res, open(OUT, "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
