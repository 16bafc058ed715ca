#!/bin/bash
# development: SQ counters of config 4's kernels (two passes; --kernel-trace + --pmc only).  usage (GPU box): bash tools/exp/pmc_capm.sh
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/pmc_capm; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $OUT/a -o a -- python $ROOT/bench.py --no-cpu --workload capm --steps 1 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/b -o b -- python $ROOT/bench.py --no-cpu --workload capm --steps 1 --warmup 1 > $OUT/b.log 2>&1
python - <<P
import sqlite3, glob
for tag in "ab":
    dbs = glob.glob("$OUT/%s/**/*.db" % tag, recursive=True)
    if not dbs: print(tag, "no db"); continue
    c = sqlite3.connect(dbs[0])
    rows = {}
    for name, cn, val in c.execute("select kernel_name, counter_name, value from counters_collection"):
        if "k_mutate_reg" in name or "k_prepare" in name: rows.setdefault((name.split("(")[0][:40], cn), []).append(val)
    for (k, cn), v in sorted(rows.items()):
        v = sorted(x for x in v if x > 0)
        if v: print(tag, k, cn, "median", v[len(v)//2], "n", len(v))
P
rm -rf $OUT/a $OUT/b
