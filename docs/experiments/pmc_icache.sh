#!/bin/bash
# development: instruction-cache counters of config 2's segment kernel (141 KB of code, 64 KB of instruction cache per CU pair).
# usage (GPU box): bash tools/exp/pmc_icache.sh
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/pmc_icache; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_INST[A-Z_]*\|SQ_INST_CYCLES[A-Z_]*\|SQC_ICACHE_INPUT[A-Z_]*" | sort -u | tr '\n' ' '; echo
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE -d $OUT/a -o a -- python $ROOT/bench.py --no-cpu --steps 1 --warmup 1 > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU -d $OUT/b -o b -- python $ROOT/bench.py --no-cpu --steps 1 --warmup 1 > $OUT/b.log 2>&1
python - <<P
import sqlite3, glob
for tag in "ab":
    dbs = glob.glob("$OUT/%s/**/*.db" % tag, recursive=True)
    if not dbs: print(tag, "no db"); print(open("$OUT/%s.log" % tag).read()[-600:]); continue
    c = sqlite3.connect(dbs[0])
    rows = {}
    for name, cn, val in c.execute("select kernel_name, counter_name, value from counters_collection"):
        if "k3_segment" in name or "k2_correct" in name: rows.setdefault((name.split("(")[0][:40], cn), []).append(val)
    for (k, cn), v in sorted(rows.items()):
        v = sorted(x for x in v if x > 0)
        if v: print(tag, k, cn, "median", v[len(v)//2], "max", v[-1], "n", len(v))
P
rm -rf $OUT/a $OUT/b
