#!/usr/bin/env python
"""Per-phase static instruction census of the register mutation kernel (gfx950 ISA of `hipcc -S -DSMCMI_ISA_MARKS`): the kernel's phase
stamps (SMCMI_PROF slots in csrc/kernels.hpp k_mutate_reg) become comments in the ISA and the instructions between two of them are
counted by class.  One MH proposal executes every phase between marks 3 and 8 once; the prologue (0-3) and epilogue (8-9) once per launch.
usage: python profiles/isa_phases.py [out.json]      (no GPU needed)"""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "smc.jl_amd", "csrc", "smcmi.hip")
OUT = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r04_isa_k_mutate_reg.json")
asm = "/tmp/smcmi_isa_marks.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-mllvm", "-disable-machine-licm", "-DSMCMI_ISA_MARKS", "-S",
                       "--cuda-device-only", "-o", asm, SRC], stderr=subprocess.DEVNULL)
lines = open(asm).read().split("\n")
PHASES = {0: "prologue: staging of the proposal / model constants (once per launch)", 1: "prologue: particle loads", 2: "prologue: rest",
          3: "Philox: MH uniform, mixture uniform, the block's 5 pairs (7 calls of Philox4x32-10)", 31: "Box-Muller: 5 x (log, sqrt, sincospi) + products",
          4: "sum of squares of the draw, density check", 5: "proposal x + L z (D x D sweep)", 6: "bounds, log-prior, log-likelihood",
          7: "exp, decision, roll-back", 8: "epilogue: stores, energy power sums, block reductions (once per launch)"}


def census(key):
    start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN5smcmi\d+" + re.escape(key) + r".*:\s*(;.*)?$", l))
    end = next(i for i in range(start, len(lines)) if ".Lfunc_end" in lines[i] and lines[i].strip().endswith(":"))
    cur, out = None, collections.OrderedDict()
    for l in lines[start + 1:end]:
        t = l.strip()
        m = re.match(r"; SMCMI_MARK (\d+)", t)
        if m:
            cur = int(m.group(1))
            out.setdefault(cur, collections.Counter())
            continue
        if cur is None or not t or t.startswith((";", ".")) or t.endswith(":"):
            continue
        op = t.split()[0]
        c = out[cur]
        c["all"] += 1
        if op.startswith("v_") and "f64" in op:
            c["valu_f64"] += 1
        elif op.startswith(("v_mad_u64", "v_mul_hi", "v_mul_lo")):
            c["valu_int_mul"] += 1
        elif op.startswith("v_"):
            c["valu_other"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith(("global", "buffer", "flat", "scratch")):
            c["vmem"] += 1
    return out


res = {}
for nice, key in (("k_mutate_reg<10,true>", "k_mutate_regILi10ELb1E"), ("k_mutate_reg<9,true>", "k_mutate_regILi9ELb1E")):
    ph = census(key)
    tot = sum(c["all"] for c in ph.values())
    valu = sum(c["valu_f64"] + c["valu_int_mul"] + c["valu_other"] for c in ph.values())
    res[nice] = {"instructions": tot, "valu": valu,
                 "phases": [{"mark": k, "what": PHASES.get(k, "?"), **{kk: int(v) for kk, v in c.items()},
                             "share_of_valu": round((c["valu_f64"] + c["valu_int_mul"] + c["valu_other"]) / max(valu, 1), 3)} for k, c in ph.items()]}
res["note"] = ("static counts between the kernel's phase marks (instructions AFTER mark k up to the next mark; loops counted once: the MH loop body, marks 3 .. 7, "
               "runs once per proposal; v_mad_u64_u32 - Philox's products - issue at a quarter of the rate of an FP64 FMA).  With the marks in, "
               "the scheduler cannot move instructions across phase boundaries: the totals are a few per cent above the production kernel's.")
json.dump(res, open(OUT, "w"), indent=1)
for k, v in res.items():
    if k == "note":
        continue
    print(k, "instructions", v["instructions"], "valu", v["valu"])
    for p in v["phases"]:
        print("   mark %2d %-70s all %5d f64 %4d intmul %3d other %4d salu %4d lds %3d vmem %3d  valu share %.3f" % (
            p["mark"], p["what"][:70], p.get("all", 0), p.get("valu_f64", 0), p.get("valu_int_mul", 0), p.get("valu_other", 0), p.get("salu", 0), p.get("lds", 0), p.get("vmem", 0), p["share_of_valu"]))
