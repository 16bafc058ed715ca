#!/bin/bash
# per-kernel table of one bench configuration.  usage: bash tools/prof_n.sh <tag> [bench args]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python $ROOT/bench.py --no-cpu "$@" > $OUT/bench.json 2> $OUT/bench.err
cd $ROOT
python profiles/summarize_rocpd.py $(find $OUT/kt -name "*.db" | head -1) > $OUT/kernel_stats.txt
rm -rf $OUT/kt
head -12 $OUT/kernel_stats.txt | cut -c1-60,72-140
